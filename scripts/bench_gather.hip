// Microbenchmark: cost of random 8-byte gathers vs the size of the gathered vector and the
// load flavour, next to the pure streaming rate of the (idx,val) arrays.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>  // 0 no gather, 1 plain, 2 nt, 3 sc1 (agent-scope relaxed atomic load)
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, const double* __restrict__ val,
                                         const double* __restrict__ x, double* out, long nnz) {
    const long per = 2048;
    double acc = 0;
    for (long base = (long)blockIdx.x * per; base < nnz; base += (long)gridDim.x * per) {
        int c[8]; double v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            long p = base + e * 256 + threadIdx.x;
            c[e] = __builtin_nontemporal_load(idx + p);
            v[e] = __builtin_nontemporal_load(val + p);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            double xg;
            if (MODE == 0) xg = (double)c[e];
            else if (MODE == 1) xg = x[c[e]];
            else if (MODE == 2) xg = __builtin_nontemporal_load(x + c[e]);
            else xg = __hip_atomic_load(x + c[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += xg * v[e];
        }
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}


// persistent workgroups walking phases (the access pattern of spmv_phased_kernel without its
// LDS staging / barriers): entries are laid out [phase][workgroup][entries]
template <int PER>
__global__ __launch_bounds__(256) void kp(const int* __restrict__ idx, const double* __restrict__ val,
                                          const double* __restrict__ x, double* out, int P, long per_step) {
    double acc = 0;
    const int G = gridDim.x, w = blockIdx.x;
    for (int p = 0; p < P; p++) {
        const long base = ((long)p * G + w) * per_step;
        for (long c = 0; c < per_step; c += 256 * PER) {
            int ci[PER]; double v[PER];
#pragma unroll
            for (int e = 0; e < PER; e++) {
                long q = base + c + e * 256 + threadIdx.x;
                ci[e] = __builtin_nontemporal_load(idx + q);
                v[e] = __builtin_nontemporal_load(val + q);
            }
#pragma unroll
            for (int e = 0; e < PER; e++) acc += x[ci[e]] * v[e];
        }
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const long nnz = 16L << 20;
    std::vector<int> hidx(nnz);
    std::vector<double> hval(nnz, 1.0);
    int* idx; double *val, *x, *out;
    CHECK(hipMalloc(&idx, nnz * 4)); CHECK(hipMalloc(&val, nnz * 8));
    CHECK(hipMalloc(&x, 64L << 20)); CHECK(hipMalloc(&out, 2048L * 256 * 8));
    CHECK(hipMemset(x, 0, 64L << 20));
    CHECK(hipMemcpy(val, hval.data(), nnz * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::mt19937_64 rng(1);
    const long sizes[] = {16L<<10, 128L<<10, 1L<<20, 2L<<20, 4L<<20, 8L<<20, 16L<<20, 64L<<20};
    for (long S : sizes) {
        const long ne = S / 8;
        for (long p = 0; p < nnz; p++) hidx[p] = (int)(rng() % ne);
        CHECK(hipMemcpy(idx, hidx.data(), nnz * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 4; mode++) {
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, idx, val, x, out, nnz);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, idx, val, x, out, nnz);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, idx, val, x, out, nnz);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, idx, val, x, out, nnz);
            };
            for (int w = 0; w < 3; w++) launch();
            CHECK(hipEventRecord(e0));
            const int reps = 20;
            for (int r = 0; r < reps; r++) launch();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms / reps * 1e3;
            printf("x=%6ld KB mode=%d: %7.1f us  stream %.2f TB/s  %.1f Ggather/s\n", S >> 10, mode, us,
                   nnz * 12.0 / us / 1e6, nnz / us / 1e3);
        }
    }
    // XCD-sliced gather: block b (XCD b % 8 under round-robin dispatch) gathers only from slice b % nsl
    for (long total : {8L<<20, 16L<<20}) for (int nsl : {4, 8}) {
        const long ne = total / 8 / nsl;
        const int grid = 2048;
        for (long p = 0; p < nnz; p++) { long chunk = p / 2048; int b = (int)(chunk % grid); hidx[p] = (int)((b % nsl) * ne + rng() % ne); }
        CHECK(hipMemcpy(idx, hidx.data(), nnz * 4, hipMemcpyHostToDevice));
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, idx, val, x, out, nnz);
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, idx, val, x, out, nnz);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("xcdsliced x=%ld MB slices=%d (%ld KB each): %.1f us  (%.1f Ggather/s)\n", total >> 20, nsl, total / nsl >> 10, ms / 20 * 1e3, nnz / (ms / 20 * 1e3) / 1e3);
    }
    // persistent-workgroup phase walk
    for (long total : {8L<<20, 16L<<20}) for (long slice : {1L<<20}) for (int G : {1024, 1280, 2048}) for (int PER : {4, 8}) {
        const int P = (int)(total / slice);
        long per_step = nnz / ((long)P * G);
        per_step = per_step / (256 * PER) * (256 * PER);
        const long ne = slice / 8;
        const long used = per_step * P * G;
        for (long q = 0; q < used; q++) { long p = q / (per_step * G); hidx[q] = (int)(p * ne + rng() % ne); }
        CHECK(hipMemcpy(idx, hidx.data(), nnz * 4, hipMemcpyHostToDevice));
        auto launch = [&]() {
            if (PER == 4) hipLaunchKernelGGL(kp<4>, dim3(G), dim3(256), 0, 0, idx, val, x, out, P, per_step);
            else hipLaunchKernelGGL(kp<8>, dim3(G), dim3(256), 0, 0, idx, val, x, out, P, per_step);
        };
        for (int w = 0; w < 3; w++) launch();
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; r++) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("phasewalk x=%ld MB P=%d G=%d PER=%d nnz=%ld: %.1f us  (%.1f Ggather/s)\n", total >> 20, P, G, PER, used, ms / 20 * 1e3, used / (ms / 20 * 1e3) / 1e3);
    }
    // time-tiled: nnz region r gathers only from slice r of a 16 MB / 8 MB vector
    for (long total : {8L<<20, 16L<<20}) for (long slice : {512L<<10, 1L<<20, 2L<<20, 4L<<20}) {
        const long nsl = total / slice, per = nnz / nsl, ne = slice / 8;
        for (long p = 0; p < nnz; p++) { long r = p / per; if (r >= nsl) r = nsl - 1; hidx[p] = (int)(r * ne + rng() % ne); }
        CHECK(hipMemcpy(idx, hidx.data(), nnz * 4, hipMemcpyHostToDevice));
        for (int grid : {1536, 2048}) {
            for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, idx, val, x, out, nnz);
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, idx, val, x, out, nnz);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("tiled x=%ld MB slice=%ld KB grid=%d: %.1f us\n", total >> 20, slice >> 10, grid, ms / 20 * 1e3);
        }
    }
    return 0;
}
