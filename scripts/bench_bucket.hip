// Microbenchmark for the two-kernel "bucket" SpMV: expand (source block resident in LDS, products
// streamed out in bin order) + reduce (one bin of products resident in LDS, per-output sums in
// storage order).  Random sparsity, sizes of the C3 passes.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>
#include <numeric>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int T = 512;

template <int U, int MODE>
__global__ __launch_bounds__(T) void expand(const double* __restrict__ x, const double* __restrict__ val,
                                            const uint16_t* __restrict__ idx, const uint32_t* __restrict__ pos,
                                            const long* __restrict__ wg_lo, const long* __restrict__ wg_hi,
                                            const int* __restrict__ wg_block, int SB, double* __restrict__ prod) {
    extern __shared__ double xs[];
    const int w = blockIdx.x;
    const double* xb = x + (long)wg_block[w] * SB;
    for (int i = threadIdx.x; i < SB; i += T) xs[i] = xb[i];
    __syncthreads();
    const long lo = wg_lo[w], hi = wg_hi[w];
    for (long base = lo; base < hi; base += (long)T * U) {
        double v[U]; uint16_t c[U]; uint32_t p[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long q = base + u * T + threadIdx.x;
            const bool ok = q < hi;
            v[u] = ok ? __builtin_nontemporal_load(val + q) : 0.0;
            c[u] = ok ? __builtin_nontemporal_load(idx + q) : 0;
            p[u] = ok ? (MODE == 1 ? (uint32_t)q : __builtin_nontemporal_load(pos + q)) : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (p[u] != 0xffffffffu) {
                if (MODE == 2) __builtin_nontemporal_store(v[u] * xs[c[u]], prod + p[u]);
                else prod[p[u]] = v[u] * xs[c[u]];
            }
    }
}

__global__ __launch_bounds__(T) void reduce(const double* __restrict__ prod, const long* __restrict__ binptr,
                                            const long* __restrict__ gptr, const uint8_t* __restrict__ dlen,
                                            const uint16_t* __restrict__ perm, const double* __restrict__ W,
                                            int DB, double* __restrict__ out) {
    extern __shared__ double ps[];
    const int b = blockIdx.x;
    const long lo = binptr[b];
    const int cnt = (int)(binptr[b + 1] - lo);
    for (int i = threadIdx.x; i < cnt; i += T) ps[i] = __builtin_nontemporal_load(prod + lo + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g = wave; g < DB / 64; g += T / 64) {
        const long d = (long)b * DB + g * 64 + lane;
        const int len = dlen[d];
        const uint16_t* pp = perm + gptr[(long)b * (DB / 64) + g] + lane;
        double s = 0;
        for (int k = 0; k < len; k++) s += ps[pp[k * 64]];
        out[d] = s * W[d];
    }
}

int main(int argc, char** argv) {
    const long nnz = 16L << 20;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Cfg { long X, D; int per, SB, DB, split; };
    std::vector<Cfg> cfgs = {
        {1L << 20, 2L << 20, 8, 8192, 1024, 8},   // pass 1: gather y (1M) into 2M columns
        {1L << 20, 2L << 20, 8, 8192, 512, 8},
        {1L << 20, 2L << 20, 8, 4096, 1024, 4},
        {2L << 20, 1L << 20, 16, 8192, 512, 4},   // pass 2: gather t (2M) into 1M rows
        {2L << 20, 1L << 20, 16, 8192, 256, 4},
        {2L << 20, 1L << 20, 16, 4096, 512, 2},
    };
    std::mt19937_64 rng(7);
    for (const Cfg& c : cfgs) {
        const long D = c.D, X = c.X; const int SB = c.SB, DB = c.DB;
        const int S = (int)(X / SB); const long NB = D / DB;
        std::vector<int> src(nnz); std::vector<double> val(nnz), x(X), W(D);
        for (auto& v : x) v = (double)(rng() % 1000) / 37.0 - 11.0;
        for (auto& v : W) v = (double)(rng() % 1000) / 91.0 + 0.1;
        for (long d = 0; d < D; d++) {
            for (int k = 0; k < c.per; k++) { src[d * c.per + k] = (int)(rng() % X); val[d * c.per + k] = (double)(rng() % 2001) / 1000.0 - 1.0; }
            std::sort(src.begin() + d * c.per, src.begin() + (d + 1) * c.per);
        }
        auto dst = [&](long e) { return e / c.per; };
        // bin order: (bin, sblock, e);  expand order: (sblock, bin, e)
        std::vector<uint64_t> key(nnz);
        std::vector<uint32_t> order(nnz), binpos(nnz);
        std::iota(order.begin(), order.end(), 0u);
        for (long e = 0; e < nnz; e++) key[e] = ((uint64_t)(dst(e) / DB) << 40) | ((uint64_t)(src[e] / SB) << 28);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        for (long i = 0; i < nnz; i++) binpos[order[i]] = (uint32_t)i;
        std::vector<long> binptr(NB + 1, 0);
        for (long e = 0; e < nnz; e++) binptr[dst(e) / DB + 1]++;
        for (long b = 0; b < NB; b++) binptr[b + 1] += binptr[b];
        long maxbin = 0; for (long b = 0; b < NB; b++) maxbin = std::max(maxbin, binptr[b + 1] - binptr[b]);
        for (long e = 0; e < nnz; e++) key[e] = ((uint64_t)(src[e] / SB) << 40) | (uint64_t)(dst(e) / DB);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        std::vector<double> xval(nnz); std::vector<uint16_t> xidx(nnz); std::vector<uint32_t> xpos(nnz);
        std::vector<long> sptr(S + 1, 0);
        for (long i = 0; i < nnz; i++) {
            const uint32_t e = order[i];
            xval[i] = val[e]; xidx[i] = (uint16_t)(src[e] % SB); xpos[i] = binpos[e]; sptr[src[e] / SB + 1]++;
        }
        for (int s = 0; s < S; s++) sptr[s + 1] += sptr[s];
        const int G = S * c.split;
        std::vector<long> wlo(G), whi(G); std::vector<int> wblk(G);
        for (int s = 0; s < S; s++) for (int k = 0; k < c.split; k++) {
            const long n = sptr[s + 1] - sptr[s];
            long piece = (n + c.split - 1) / c.split; piece = (piece + 63) / 64 * 64;
            wlo[s * c.split + k] = std::min(sptr[s] + k * piece, sptr[s + 1]);
            whi[s * c.split + k] = std::min(sptr[s] + (k + 1) * piece, sptr[s + 1]);
            wblk[s * c.split + k] = s;
        }
        // reduce metadata
        const long NG = D / 64;
        std::vector<long> gptr(NG + 1, 0); std::vector<uint8_t> dlen(D, (uint8_t)c.per);
        for (long g = 0; g < NG; g++) gptr[g + 1] = gptr[g] + 64L * c.per;
        std::vector<uint16_t> perm(gptr[NG]);
        for (long d = 0; d < D; d++) for (int k = 0; k < c.per; k++) {
            const long e = d * c.per + k;
            perm[gptr[d / 64] + k * 64 + d % 64] = (uint16_t)(binpos[e] - binptr[d / DB]);
        }
        std::vector<double> ref(D);
        for (long d = 0; d < D; d++) { double s = 0; for (int k = 0; k < c.per; k++) s += val[d * c.per + k] * x[src[d * c.per + k]]; ref[d] = s * W[d]; }

        double *dx, *dval, *dprod, *dW, *dout; uint16_t *didx, *dperm; uint32_t* dpos; long *dwlo, *dwhi, *dbinptr, *dgptr; int* dwblk; uint8_t* ddlen;
        CHECK(hipMalloc(&dx, X * 8)); CHECK(hipMalloc(&dval, nnz * 8)); CHECK(hipMalloc(&dprod, nnz * 8)); CHECK(hipMalloc(&dW, D * 8)); CHECK(hipMalloc(&dout, D * 8));
        CHECK(hipMalloc(&didx, nnz * 2)); CHECK(hipMalloc(&dperm, perm.size() * 2)); CHECK(hipMalloc(&dpos, nnz * 4));
        CHECK(hipMalloc(&dwlo, G * 8)); CHECK(hipMalloc(&dwhi, G * 8)); CHECK(hipMalloc(&dwblk, G * 4));
        CHECK(hipMalloc(&dbinptr, (NB + 1) * 8)); CHECK(hipMalloc(&dgptr, (NG + 1) * 8)); CHECK(hipMalloc(&ddlen, D));
        CHECK(hipMemcpy(dx, x.data(), X * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dval, xval.data(), nnz * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dW, W.data(), D * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(didx, xidx.data(), nnz * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dperm, perm.data(), perm.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dpos, xpos.data(), nnz * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dwlo, wlo.data(), G * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dwhi, whi.data(), G * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dwblk, wblk.data(), G * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dbinptr, binptr.data(), (NB + 1) * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dgptr, gptr.data(), (NG + 1) * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(ddlen, dlen.data(), D, hipMemcpyHostToDevice));
        CHECK(hipMemset(dprod, 0xff, nnz * 8));
        const size_t lds_e = (size_t)SB * 8, lds_r = (size_t)maxbin * 8;
        CHECK(hipFuncSetAttribute((const void*)expand<4,0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CHECK(hipFuncSetAttribute((const void*)expand<4,1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CHECK(hipFuncSetAttribute((const void*)expand<4,2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CHECK(hipFuncSetAttribute((const void*)reduce, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        auto run_e = [&](int U) {
            if (U == 4) hipLaunchKernelGGL((expand<4,0>), dim3(G), dim3(T), lds_e, 0, dx, dval, didx, dpos, dwlo, dwhi, dwblk, SB, dprod);
            else if (U == 8) hipLaunchKernelGGL((expand<4,1>), dim3(G), dim3(T), lds_e, 0, dx, dval, didx, dpos, dwlo, dwhi, dwblk, SB, dprod);
            else hipLaunchKernelGGL((expand<4,2>), dim3(G), dim3(T), lds_e, 0, dx, dval, didx, dpos, dwlo, dwhi, dwblk, SB, dprod);
        };
        auto run_r = [&]() { hipLaunchKernelGGL(reduce, dim3((int)NB), dim3(T), lds_r, 0, dprod, dbinptr, dgptr, ddlen, dperm, dW, DB, dout); };
        run_e(4); run_r(); CHECK(hipDeviceSynchronize());
        std::vector<double> got(D);
        CHECK(hipMemcpy(got.data(), dout, D * 8, hipMemcpyDeviceToHost));
        long bad = 0; for (long d = 0; d < D; d++) bad += got[d] != ref[d];
        auto timeit = [&](auto&& f) {
            for (int w = 0; w < 3; w++) f();
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 20; r++) f();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / 20 * 1e3;
        };
        const double te4 = timeit([&] { run_e(4); }), te8 = timeit([&] { run_e(8); }), te2 = timeit([&] { run_e(2); }), tr = timeit(run_r);
        const double both = timeit([&] { run_e(4); run_r(); });
        printf("X=%ldK D=%ldK SB=%d DB=%d split=%d G=%d NB=%ld maxbin=%ld mismatches=%ld: expand %.1f us seqwrite %.1f us ntstore %.1f (%.2f TB/s)  reduce %.1f us (%.2f TB/s)  pair %.1f us\n",
               X >> 10, D >> 10, SB, DB, c.split, G, NB, maxbin, bad, te4, te8, te2, nnz * 22.0 / std::min(te4, te8) / 1e6, tr, nnz * 10.0 / tr / 1e6, both);
        hipFree(dx); hipFree(dval); hipFree(dprod); hipFree(dW); hipFree(dout); hipFree(didx); hipFree(dperm); hipFree(dpos);
        hipFree(dwlo); hipFree(dwhi); hipFree(dwblk); hipFree(dbinptr); hipFree(dgptr); hipFree(ddlen);
    }
    return 0;
}
