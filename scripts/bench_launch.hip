// Microbenchmark: cost of a dependent chain of small kernels on one stream (gfx950), as a floor
// for level-scheduled triangular solves: empty kernel, and kernels with 1/2/3 dependent loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int DEPTH>
__global__ void chain(const int* __restrict__ a, const int* __restrict__ b, double* x, int n) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (DEPTH == 0) return;
    int i = a[k];
    if (DEPTH >= 2) i = b[i];
    double v = x[i];
    if (DEPTH >= 3) v += x[(i * 7 + 1) % n];
    x[k] = v * 0.5 + 1.0;
}
int main() {
    const int n = 1 << 16;
    std::vector<int> h(n);
    for (int i = 0; i < n; i++) h[i] = (i * 9973) % n;
    int *a, *b; double* x;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&x, n * 8));
    CHECK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(x, 0, n * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rows : {256, 8192, 65536}) for (int depth = 0; depth < 4; depth++) {
        auto launch = [&]() {
            dim3 g((rows + 255) / 256), t(256);
            if (depth == 0) hipLaunchKernelGGL(chain<0>, g, t, 0, 0, a, b, x, rows);
            if (depth == 1) hipLaunchKernelGGL(chain<1>, g, t, 0, 0, a, b, x, rows);
            if (depth == 2) hipLaunchKernelGGL(chain<2>, g, t, 0, 0, a, b, x, rows);
            if (depth == 3) hipLaunchKernelGGL(chain<3>, g, t, 0, 0, a, b, x, rows);
        };
        for (int w = 0; w < 50; w++) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 2000;
        for (int r = 0; r < reps; r++) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("rows %6d depth %d: %.2f us per dependent launch\n", rows, depth, ms / reps * 1e3);
    }
    return 0;
}
