/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the CPU kernels the reference spends its
 * Krylov iterations in, used by bench.py's cpu_baseline leg (kind "port") and by tests as a checker.
 *
 *   csr_spmv    = PETSc MatMult_SeqAIJ on the MATIS local matrix (called at geneo.cpp:1931,1935,
 *                 driver:831,1075 and by KSP at driver:1240): y = A x, FP64 values, 32-bit indices.
 *   dot / axpy  = VecDot / VecAXPY of the Krylov loop.
 * OpenMP over rows (the reference runs one MPI rank per core; threads stand in for ranks here).
 * Never linked into the product. */
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oracle_csr_spmv(int n, const int* rowptr, const int* col, const double* val, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * x[col[k]];
    y[i] = s;
  }
}

double oracle_dot(int n, const double* x, const double* y) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

void oracle_axpy(int n, double a, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) y[i] += a * x[i];
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
