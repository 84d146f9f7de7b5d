/* TEST INFRASTRUCTURE ONLY (part of the oracle/_ref recipe).
 *
 * The reference calls LAPACK through the Fortran names dpotrf_/dpotrs_/dtrcon_
 * (reference src/lapack.cc:16-23).  The image has LAPACK only inside scipy's
 * bundled OpenBLAS, which exports the same routines with a "scipy_" prefix
 * (LP64 interface).  These three functions forward the calls unchanged; they
 * contain no arithmetic.
 */
extern void scipy_dpotrf_(const char*, const int*, double*, const int*, int*);
extern void scipy_dpotrs_(const char*, const int*, const int*, const double*,
                          const int*, double*, const int*, int*);
extern void scipy_dtrcon_(const char*, const char*, const char*, const int*,
                          const double*, const int*, double*, double*, int*,
                          int*);

void dpotrf_(const char* uplo, const int* n, double* a, const int* lda,
             int* info) {
    scipy_dpotrf_(uplo, n, a, lda, info);
}
void dpotrs_(const char* uplo, const int* n, const int* nrhs, const double* a,
             const int* lda, double* b, const int* ldb, int* info) {
    scipy_dpotrs_(uplo, n, nrhs, a, lda, b, ldb, info);
}
void dtrcon_(const char* norm, const char* uplo, const char* diag,
             const int* n, const double* a, const int* lda, double* rcond,
             double* work, int* iwork, int* info) {
    scipy_dtrcon_(norm, uplo, diag, n, a, lda, rcond, work, iwork, info);
}
