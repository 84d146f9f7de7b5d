// Microbenchmark: cost of a grid-wide barrier inside one cooperative launch on MI355X (agent-scope
// atomic counter + polling), with and without a dependent agent-scope store/load per round -- the
// hand-off a single-launch level-scheduled triangular sweep would need.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void rounds_kernel(unsigned* counter, int* abort_flag, double* x, int rounds, int exchange) {
    const unsigned G = gridDim.x;
    double v = 1.0;
    for (int r = 0; r < rounds; r++) {
        if (exchange) {
            // publish one value per thread, consume a value written by another workgroup last round
            const size_t mine = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
            __hip_atomic_store(x + mine, v + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * G;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (exchange) {
            const size_t other = ((size_t)((blockIdx.x + 37) % G)) * blockDim.x + threadIdx.x;
            v = __hip_atomic_load(x + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (v == -1.0) x[0] = v;
}

int main() {
    unsigned* counter; int* abort_flag; double* x;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&abort_flag, 4)); CHECK(hipMalloc(&x, 2048 * 256 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rounds_kernel, 256, 0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d, resident blocks per CU %d\n", prop.multiProcessorCount, per_cu);
    for (int G : {16, 32, 64, 128, 256}) for (int exchange : {0, 1}) {
        if (G > per_cu * prop.multiProcessorCount) continue;
        int rounds = 200;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(abort_flag, 0, 4));
            void* args[] = {&counter, &abort_flag, &x, &rounds, &exchange};
            CHECK(hipEventRecord(e0));
            CHECK(hipLaunchCooperativeKernel((const void*)rounds_kernel, dim3(G), dim3(256), args, 0, 0));
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            int ab; CHECK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
            if (rep == 1) printf("G=%4d exchange=%d: %.2f us per round%s\n", G, exchange, ms * 1e3 / rounds, ab ? "  (ABORTED)" : "");
        }
    }
    return 0;
}
