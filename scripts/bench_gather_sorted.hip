// Microbenchmark (round 3): does ordering the gathers of a tile by ADDRESS pay?  XCD-sliced random 8-byte gathers
// (block b gathers only from slice b % nsl, like spmv_sliced_tile_kernel), entries of each tile of C entries sorted
// by gathered index, so that lanes of one instruction (and consecutive instructions of one CU) that fall on the same
// 128-byte line share one L1->L2 request.  C = 2048 is the current tile; larger C = more entries per line.
// Also: the same with the products written to a permuted LDS slot (what a row-ordered reduction needs).  gfx950.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// tile t (C entries) is handled by block t % grid; inside a tile, batches of 2048 entries: 8 per thread
template <bool LDS>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ idx, const double* __restrict__ val,
                                         const double* __restrict__ x, double* out, long ntiles, int C, int slice_elems, int nsl) {
    extern __shared__ double prod[];
    double acc = 0;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const double* xs = x + (size_t)(t % nsl) * slice_elems;
        const long e0 = t * C;
        for (int base = 0; base < C; base += 2048) {
            unsigned c[8]; double v[8], xg[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const long p = e0 + base + e * 256 + threadIdx.x;
                c[e] = __builtin_nontemporal_load(idx + p);
                v[e] = __builtin_nontemporal_load(val + p);
            }
#pragma unroll
            for (int e = 0; e < 8; e++) xg[e] = xs[c[e] & 0x3ffff];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (LDS) prod[c[e] >> 18] = xg[e] * v[e];
                else acc += xg[e] * v[e];
            }
        }
        if (LDS) {
            __syncthreads();
            for (int i = threadIdx.x; i < C; i += 256) acc += prod[i];
            __syncthreads();
        }
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const long nnz = 16L << 20;
    std::vector<unsigned> hidx(nnz);
    std::vector<double> hval(nnz, 1.0);
    unsigned* idx; double *val, *x, *out;
    CHECK(hipMalloc(&idx, nnz * 4)); CHECK(hipMalloc(&val, nnz * 8));
    CHECK(hipMalloc(&x, 64L << 20)); CHECK(hipMalloc(&out, 4096L * 256 * 8));
    CHECK(hipMemset(x, 0, 64L << 20));
    CHECK(hipMemcpy(val, hval.data(), nnz * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::mt19937_64 rng(1);
    for (long total : {8L << 20, 16L << 20}) for (long slice : {2L << 20, 1L << 20}) {
        const int nsl = (int)(total / slice);
        const int ne = (int)(slice / 8);
        for (int C : {2048, 4096, 8192, 16384}) for (int sorted = 0; sorted < 2; sorted++) for (int lds = 0; lds < 2; lds++) {
            const long ntiles = nnz / C;
            for (long t = 0; t < ntiles; t++) {
                std::vector<unsigned> a(C);
                for (int i = 0; i < C; i++) a[i] = (unsigned)(rng() % ne);
                if (sorted) std::sort(a.begin(), a.end());
                // LDS slot: a random permutation of the tile (row order is unrelated to address order)
                std::vector<unsigned> perm(C);
                for (int i = 0; i < C; i++) perm[i] = i;
                if (sorted) std::shuffle(perm.begin(), perm.end(), rng);
                for (int i = 0; i < C; i++) hidx[t * C + i] = a[i] | (perm[i] << 18);
            }
            CHECK(hipMemcpy(idx, hidx.data(), nnz * 4, hipMemcpyHostToDevice));
            // grid: a multiple of 8 * nsl so that tile t -> slice t % nsl and block -> XCD block % 8 stay aligned
            const int grid = 2048;
            const size_t shm = lds ? (size_t)C * 8 : 0;
            auto launch = [&]() {
                if (lds) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), shm, 0, idx, val, x, out, ntiles, C, ne, nsl);
                else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, idx, val, x, out, ntiles, C, ne, nsl);
            };
            for (int w = 0; w < 3; w++) launch();
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 20; r++) launch();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            // expected distinct lines per tile
            const double L = slice / 128.0;
            const double distinct = L * (1.0 - std::exp(-(double)C / L));
            printf("x=%2ld MB slice=%4ld KB tile=%5d %s %s: %6.1f us  (%.1f Ggather/s; lines/tile %.0f = %.2f of entries)\n", total >> 20, slice >> 10, C,
                   sorted ? "sorted  " : "unsorted", lds ? "lds-perm" : "reg-acc ", ms / 20 * 1e3, nnz / (ms / 20 * 1e3) / 1e3, distinct, distinct / C);
            fflush(stdout);
        }
    }
    return 0;
}
