// Microbenchmark (round 4): a tile kernel for the XCD-sliced gather SpMV that keeps the ROW SUMS of a row block in LDS
// (ds_add_f64) instead of staging every product in an LDS slot and summing rows afterwards.
//   * tile (rb, s) = the entries of RB rows whose gathered index lies in slice s, sorted by gathered ADDRESS over the
//     whole slice (no sub-slices: nothing but the RB accumulators has to fit LDS, so RB = 16384 rows -> 2 entries
//     per 128-byte line of a 2 MiB slice at C3), cut into batches of T*8 entries with ONE barrier per batch;
//   * the layout guarantees that a batch holds at most one entry of a row (an entry that would be the second one is
//     deferred to the next batch), so the adds of a batch hit distinct addresses and a row is summed in batch order =
//     ascending address order: deterministic, and equal to the storage order when rows are stored with ascending
//     indices;
//   * variants: partial vectors + separate combine launch (today's scheme), or the LAST workgroup of a row block to
//     finish (agent-scope counter) folds the other slices' partials in slice order and runs the epilogue itself.
// Synthetic C3-shaped passes: pass 1 = 2M rows x 8 entries gathering from 1M doubles (4 slices), pass 2 = 1M rows x 16
// entries gathering from 2M doubles (8 slices).  gfx950.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

typedef unsigned long long u64;
constexpr int kOffBits = 18;

struct View {
    int nrows, nrows_pad, ns, nrb, RB, slice;
    const unsigned* tile_ptr;      // [nrb*ns+1]
    const unsigned* pack;          // row in block << 18 | offset in slice
    const double* val;
    double* partial;               // [ns][nrows_pad]
    unsigned* counter;             // [nrb] arrivals (fold)
    const double* w;               // epilogue weight (pass 1: out = acc * w[r]; pass 2: out = y[r]*w[r] + acc, dot += y[r]*out)
    const double* y;
    double* out;
    double* dot_partials;          // [nrb]
};

__device__ __forceinline__ double load_sc1(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store2_sc1(double* p, double a, double b) {      // one 16-byte write-through store
    d2_t v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_add(double* p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// MODE 0: partial vectors (plain stores); 1: fold by the last arriver (8-byte sc1 stores); 2: like 0 without the per-batch
// barrier (upper bound); 3: fold with 16-byte sc1 stores; ablations of mode 0: 4 = no gather (x = 1), 5 = no LDS add (register sum),
// 6 = no partial store, 7 = next batch's stream issued AFTER the adds (no overlap)
template <int T, int MODE, bool PASS2, int U = 8>
__global__ __launch_bounds__(T) void tile_kernel(View M, const double* __restrict__ x) {
    extern __shared__ double acc[];
    __shared__ unsigned s_old;
    __shared__ double red[T / 64];
    constexpr int B = T * U;
    constexpr int NS = PASS2 ? 8 : 4;
    const int tid = threadIdx.x;
    const int ntiles = M.nrb * M.ns;
    unsigned pk[U];
    double v[U];
    double regsum = 0.0;
    auto stream = [&](int tile, int base) {
        const unsigned e0 = M.tile_ptr[tile];
        const int ne = (int)(M.tile_ptr[tile + 1] - e0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(base + u * T + tid, max(ne - 1, 0));
            pk[u] = __builtin_nontemporal_load(M.pack + e0 + i);
            v[u] = __builtin_nontemporal_load(M.val + e0 + i);
        }
    };
    if ((int)blockIdx.x < ntiles) stream(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s = tile % M.ns, rb = tile / M.ns;
        const double* __restrict__ xs = x + (size_t)s * M.slice;
        const int ne = (int)(M.tile_ptr[tile + 1] - M.tile_ptr[tile]);
        for (int r = tid; r < M.RB; r += T) acc[r] = 0.0;
        for (int base = 0; base < ne || base == 0; base += B) {
            double xg[U], vv[U];
            unsigned row[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                row[u] = pk[u] >> kOffBits;
                vv[u] = v[u];
                xg[u] = MODE == 4 ? 1.0 : xs[pk[u] & ((1u << kOffBits) - 1u)];
            }
            if (MODE != 7) {
                if (base + B < ne) stream(tile, base + B);
                else if (tile + (int)gridDim.x < ntiles) stream(tile + gridDim.x, 0);
            }
            if (MODE != 2 || base == 0) __syncthreads();          // the adds of the previous batch (or the zeroing) are done
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + u * T + tid;
                if (MODE == 5) { if (i < ne) regsum += xg[u] * vv[u] + (double)row[u]; }
                else if (i < ne) lds_add(acc + row[u], xg[u] * vv[u]);
            }
            if (MODE == 7) {
                if (base + B < ne) stream(tile, base + B);
                else if (tile + (int)gridDim.x < ntiles) stream(tile + gridDim.x, 0);
            }
        }
        __syncthreads();
        double* mine = M.partial + (size_t)s * M.nrows_pad + (size_t)rb * M.RB;
        if (MODE == 5 && regsum == 12345.678) mine[tid] = regsum;
        if (MODE == 6) { if (acc[tid] == 12345.678) mine[tid] = acc[tid]; }
        else if (MODE != 1 && MODE != 3) {
            for (int r = tid; r < M.RB; r += T) mine[r] = acc[r];
        } else {
            if (MODE == 1) for (int r = tid; r < M.RB; r += T) store_sc1(mine + r, acc[r]);
            else for (int r = 2 * tid; r < M.RB; r += 2 * T) store2_sc1(mine + r, acc[r], acc[r + 1]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) s_old = __hip_atomic_fetch_add(M.counter + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (s_old == (unsigned)(M.ns - 1)) {       // every other slice of this row block has been stored
                double dot = 0.0;
                // the other slices' partial sums of G rows at a time: all G * NS loads in flight before the first add
                constexpr int G = 4;
                for (int r0 = 0; r0 < M.RB; r0 += G * T) {
                    double t[G][NS];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const int rr = r0 + g * T + tid;
                        const int r = min(rb * M.RB + rr, M.nrows - 1);
#pragma unroll
                        for (int q = 0; q < NS; q++) t[g][q] = load_sc1(M.partial + (size_t)q * M.nrows_pad + r);
                    }
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const int rr = r0 + g * T + tid;
                        const int r = rb * M.RB + rr;
                        if (rr < M.RB && r < M.nrows) {
                            double a = PASS2 ? M.y[r] * M.w[r] : 0.0;
#pragma unroll
                            for (int q = 0; q < NS; q++) a += q == s ? acc[rr] : t[g][q];
                            if (PASS2) { M.out[r] = a; dot += M.y[r] * a; }
                            else M.out[r] = a * M.w[r];
                        }
                    }
                }
                if (PASS2) {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
                    if ((tid & 63) == 0) red[tid >> 6] = dot;
                    __syncthreads();
                    if (tid == 0) { double t = 0.0; for (int wv = 0; wv < T / 64; wv++) t += red[wv]; M.dot_partials[rb] = t; }
                }
                if (tid == 0) __hip_atomic_store(M.counter + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();                   // acc is zeroed again by the next tile
    }
}

template <bool PASS2>
__global__ __launch_bounds__(256) void combine_kernel(View M) {
    __shared__ double red[4];
    double dot = 0.0;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < M.nrows; r += gridDim.x * 256) {
        double a = PASS2 ? M.y[r] * M.w[r] : 0.0;
        for (int q = 0; q < M.ns; q++) a += __builtin_nontemporal_load(M.partial + (size_t)q * M.nrows_pad + r);
        if (PASS2) { M.out[r] = a; dot += M.y[r] * a; }
        else M.out[r] = a * M.w[r];
    }
    if (PASS2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) M.dot_partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

struct Pass {
    int nrows, ncols, k, ns, RB, T, U;
    std::vector<unsigned> tile_ptr, pack;
    std::vector<double> val;
    std::vector<int> idx;          // plain rows (reference)
    std::vector<double> pval;
    int nrb, slice, deferred = 0;
};

static void build(Pass& P, std::mt19937_64& rng) {
    const int B = P.T * P.U;
    P.slice = ((P.ncols + P.ns - 1) / P.ns + 15) / 16 * 16;
    P.nrb = (P.nrows + P.RB - 1) / P.RB;
    const long nz = (long)P.nrows * P.k;
    P.idx.resize(nz); P.pval.resize(nz);
    std::uniform_real_distribution<double> ud(0.5, 4.0);
    for (long r = 0; r < P.nrows; r++) {
        int* row = &P.idx[r * P.k];
        for (int j = 0; j < P.k; j++) row[j] = (int)(rng() % (u64)P.ncols);
        std::sort(row, row + P.k);
        for (int j = 0; j < P.k; j++) P.pval[r * P.k + j] = (rng() & 1 ? 1.0 : -1.0) * ud(rng);
    }
    const int ntiles = P.nrb * P.ns;
    std::vector<unsigned> cnt(ntiles + 1, 0);
    for (long r = 0; r < P.nrows; r++)
        for (int j = 0; j < P.k; j++) cnt[(r / P.RB) * P.ns + P.idx[r * P.k + j] / P.slice + 1]++;
    for (int t = 0; t < ntiles; t++) cnt[t + 1] += cnt[t];
    P.tile_ptr = cnt;
    struct E { unsigned off, row; double v; };
    std::vector<E> all(nz);
    {
        std::vector<unsigned> cur(cnt.begin(), cnt.end() - 1);
        for (long r = 0; r < P.nrows; r++)
            for (int j = 0; j < P.k; j++) {
                const int c = P.idx[r * P.k + j];
                const int t = (int)(r / P.RB) * P.ns + c / P.slice;
                all[cur[t]++] = E{(unsigned)(c % P.slice), (unsigned)(r % P.RB), P.pval[r * P.k + j]};
            }
    }
    P.pack.resize(nz); P.val.resize(nz);
    std::vector<int> stamp(P.RB);
    std::vector<E> pend, nextpend, merged;
    for (int t = 0; t < ntiles; t++) {
        E* a = &all[cnt[t]];
        const int ne = (int)(cnt[t + 1] - cnt[t]);
        std::stable_sort(a, a + ne, [](const E& x, const E& y) { return x.off < y.off; });
        // batches: at most one entry of a row per batch; a second one is deferred (keeps its place in address order)
        std::fill(stamp.begin(), stamp.end(), -1);
        pend.clear();
        int cursor = 0, put = 0, batch = 0;
        while (put < ne) {
            int inb = 0;
            nextpend.clear();
            size_t pi = 0;
            while (inb < B && (pi < pend.size() || cursor < ne)) {
                E e;
                if (pi < pend.size() && (cursor >= ne || pend[pi].off <= a[cursor].off)) e = pend[pi++];
                else e = a[cursor++];
                if (stamp[e.row] == batch) { nextpend.push_back(e); P.deferred++; continue; }
                stamp[e.row] = batch;
                P.pack[cnt[t] + put] = (e.row << kOffBits) | e.off;
                P.val[cnt[t] + put] = e.v;
                put++; inb++;
            }
            for (; pi < pend.size(); pi++) nextpend.push_back(pend[pi]);
            std::sort(nextpend.begin(), nextpend.end(), [](const E& x, const E& y) { return x.off < y.off; });
            pend.swap(nextpend);
            batch++;
        }
    }
}

template <class T> static T* up(const std::vector<T>& h) {
    T* d; CHECK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

template <int T, bool PASS2, int U = 8>
static void run(Pass& P, const char* name) {
    const int nrows_pad = P.nrb * P.RB;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    std::vector<double> hx(P.ncols + 64), hw(P.nrows), hy(P.nrows);
    for (auto& t : hx) t = ud(rng);
    for (auto& t : hw) t = 0.5 + fabs(ud(rng));
    for (auto& t : hy) t = ud(rng);
    // reference
    std::vector<double> ref(P.nrows);
    double refdot = 0.0;
    for (long r = 0; r < P.nrows; r++) {
        double a = 0.0;
        for (int j = 0; j < P.k; j++) a += hx[P.idx[r * P.k + j]] * P.pval[r * P.k + j];
        if (PASS2) { a += hy[r] * hw[r]; refdot += hy[r] * a; ref[r] = a; }
        else ref[r] = a * hw[r];
    }
    View M;
    M.nrows = P.nrows; M.nrows_pad = nrows_pad; M.ns = P.ns; M.nrb = P.nrb; M.RB = P.RB; M.slice = P.slice;
    M.tile_ptr = up(P.tile_ptr); M.pack = up(P.pack); M.val = up(P.val);
    M.w = up(hw); M.y = up(hy);
    double* dx = up(hx);
    CHECK(hipMalloc(&M.partial, (size_t)P.ns * nrows_pad * 8));
    CHECK(hipMalloc(&M.counter, P.nrb * 4)); CHECK(hipMemset(M.counter, 0, P.nrb * 4));
    CHECK(hipMalloc(&M.out, (size_t)nrows_pad * 8));
    CHECK(hipMalloc(&M.dot_partials, 4096 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t lds = (size_t)P.RB * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 0, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 1, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 2, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 3, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ntiles = P.nrb * P.ns;
    auto check = [&](int ndot) {
        std::vector<double> ho(P.nrows), hd(4096);
        CHECK(hipMemcpy(ho.data(), M.out, (size_t)P.nrows * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hd.data(), M.dot_partials, 4096 * 8, hipMemcpyDeviceToHost));
        double err = 0.0, nrm = 0.0, dot = 0.0;
        for (long r = 0; r < P.nrows; r++) { err = std::max(err, fabs(ho[r] - ref[r])); nrm = std::max(nrm, fabs(ref[r])); }
        for (int i = 0; i < ndot; i++) dot += hd[i];
        printf("      check: max err %.2e (rel %.1e)%s", err, err / nrm, PASS2 ? "" : "\n");
        if (PASS2) printf(", dot rel err %.1e\n", fabs(dot - refdot) / fabs(refdot));
    };
    for (int mode : {0}) for (int grid : {ntiles}) {
        if (grid > ntiles) continue;
        if (grid != ntiles && grid % (8 * P.ns) != 0 && (8 % P.ns != 0 || grid % 8 != 0)) continue;
        CHECK(hipMemset(M.out, 0, (size_t)nrows_pad * 8));
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL((tile_kernel<T, 0, PASS2, U>), dim3(grid), dim3(T), lds, 0, M, dx);
            else if (mode == 1) hipLaunchKernelGGL((tile_kernel<T, 1, PASS2, U>), dim3(grid), dim3(T), lds, 0, M, dx);
            else if (mode == 2) hipLaunchKernelGGL((tile_kernel<T, 2, PASS2, U>), dim3(grid), dim3(T), lds, 0, M, dx);
            else hipLaunchKernelGGL((tile_kernel<T, 3, PASS2, U>), dim3(grid), dim3(T), lds, 0, M, dx);
            if (mode != 1 && mode != 3) hipLaunchKernelGGL((combine_kernel<PASS2>), dim3(1024), dim3(256), 0, 0, M);
        };
        for (int wu = 0; wu < 3; wu++) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; r++) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %s RB=%5d T=%4d U=%2d grid=%4d %-28s %6.1f us per pass\n", name, P.RB, T, U, grid,
               mode == 0 ? "partials + combine launch" : mode == 1 ? "fold by the last arriver" : mode == 2 ? "no batch barrier + combine" : "fold, 16-byte sc1 stores", ms / 20 * 1e3);
        check(mode == 1 || mode == 3 ? P.nrb : 1024);
        fflush(stdout);
    }
    // tile kernel alone: mode 0 and its ablations
    auto time_mode = [&](int mode) {
        auto launch = [&]() {
            switch (mode) {
                case 0: hipLaunchKernelGGL((tile_kernel<T, 0, PASS2, U>), dim3(ntiles), dim3(T), lds, 0, M, dx); break;
                case 4: hipLaunchKernelGGL((tile_kernel<T, 4, PASS2, U>), dim3(ntiles), dim3(T), lds, 0, M, dx); break;
                case 5: hipLaunchKernelGGL((tile_kernel<T, 5, PASS2, U>), dim3(ntiles), dim3(T), lds, 0, M, dx); break;
                case 6: hipLaunchKernelGGL((tile_kernel<T, 6, PASS2, U>), dim3(ntiles), dim3(T), lds, 0, M, dx); break;
                default: hipLaunchKernelGGL((tile_kernel<T, 7, PASS2, U>), dim3(ntiles), dim3(T), lds, 0, M, dx); break;
            }
        };
        for (int wu = 0; wu < 3; wu++) launch();
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; r++) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms / 20 * 1e3;
    };
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 4, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 5, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 6, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<T, 7, PASS2, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("  %s RB=%5d T=%4d U=%2d tile kernel alone: full %6.1f us | no gather %6.1f | no LDS add %6.1f | no partial store %6.1f | stream not overlapped %6.1f\n",
           name, P.RB, T, U, time_mode(0), time_mode(4), time_mode(5), time_mode(6), time_mode(7));
    hipFree((void*)M.tile_ptr); hipFree((void*)M.pack); hipFree((void*)M.val); hipFree((void*)M.w); hipFree((void*)M.y);
    hipFree(dx); hipFree(M.partial); hipFree(M.counter); hipFree(M.out); hipFree(M.dot_partials);
}

int main(int argc, char** argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 1;      // 1 = C3 (1M x 2M); 8 = one eighth (quick check)
    std::mt19937_64 rng(12345);
    const int m = (1 << 20) / scale, n = (2 << 20) / scale;
    struct Cfg { int RB, T, U; };
    for (Cfg c : {Cfg{16384, 512, 4}, Cfg{16384, 1024, 4}, Cfg{16384, 512, 8}}) {
        Pass P1{n, m, 8, 4, c.RB, c.T, c.U}, P2{m, n, 16, 8, c.RB, c.T, c.U};
        build(P1, rng);
        build(P2, rng);
        printf("RB=%d T=%d U=%d: pass 1 %d tiles, deferred %.2f%% of the entries; pass 2 %d tiles, deferred %.2f%%\n", c.RB, c.T, c.U, P1.nrb * P1.ns,
               100.0 * P1.deferred / ((double)P1.nrows * P1.k), P2.nrb * P2.ns, 100.0 * P2.deferred / ((double)P2.nrows * P2.k));
        if (c.T == 512 && c.U == 8) { run<512, false, 8>(P1, "pass1"); run<512, true, 8>(P2, "pass2"); }
        if (c.T == 512 && c.U == 4) { run<512, false, 4>(P1, "pass1"); run<512, true, 4>(P2, "pass2"); }
        if (c.T == 1024 && c.U == 4) { run<1024, false, 4>(P1, "pass1"); run<1024, true, 4>(P2, "pass2"); }
        if (c.T == 1024 && c.U == 8) { run<1024, false, 8>(P1, "pass1"); run<1024, true, 8>(P2, "pass2"); }
        if (c.T == 256 && c.U == 16) { run<256, false, 16>(P1, "pass1"); run<256, true, 16>(P2, "pass2"); }
    }
    return 0;
}
