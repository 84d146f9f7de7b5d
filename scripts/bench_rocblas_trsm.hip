// Microbenchmark: the inverse of a dense LU-factored block D22 = (L+I) U (kb x kb, column major, L below / U on and above the
// diagonal) as two rocblas_dtrsm on the identity -- what bump_inverse_kernel (trisolve.hip) computes with one blocked solve
// per column.  usage: bench_rocblas_trsm [kb ...]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { if ((x) != 0) { printf("%s failed\n", #x); exit(1); } } while (0)
int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {1024, 2048, 4096, 8000};
    auto t0 = std::chrono::steady_clock::now();
    rocblas_handle h;
    CHECK(rocblas_create_handle(&h));
    printf("rocblas_create_handle %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    for (int kb : sizes) {
        std::vector<double> D((size_t)kb * kb), I((size_t)kb * kb, 0.0);
        srand(1);
        for (int j = 0; j < kb; j++)
            for (int i = 0; i < kb; i++) D[(size_t)j * kb + i] = i == j ? 2.0 + (rand() % 100) * 0.01 : (rand() % 200 - 100) * 0.001;
        for (int i = 0; i < kb; i++) I[(size_t)i * kb + i] = 1.0;
        double *dD, *dX;
        CHECK(hipMalloc(&dD, sizeof(double) * kb * kb)); CHECK(hipMalloc(&dX, sizeof(double) * kb * kb));
        CHECK(hipMemcpy(dD, D.data(), sizeof(double) * kb * kb, hipMemcpyHostToDevice));
        const double one = 1.0;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemcpy(dX, I.data(), sizeof(double) * kb * kb, hipMemcpyHostToDevice));
            CHECK(hipDeviceSynchronize());
            auto a = std::chrono::steady_clock::now();
            CHECK(rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_unit, kb, kb, &one, dD, kb, dX, kb));
            CHECK(rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, kb, kb, &one, dD, kb, dX, kb));
            CHECK(hipDeviceSynchronize());
            printf("kb %5d rep %d: two dtrsm on the identity %.2f ms\n", kb, rep, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count());
        }
        (void)hipFree(dD); (void)hipFree(dX);
    }
    rocblas_destroy_handle(h);
    return 0;
}
