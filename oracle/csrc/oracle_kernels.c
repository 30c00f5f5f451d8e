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

/* NUMA placement for the timed SpMV: copies of the five arrays whose pages are first touched by the thread that will read
 * them (same static row schedule as oracle_csr_spmv; the entries of row i are touched by the owner of row i).  Arrays
 * filled by one Python thread live on one memory node and throttle a two-socket host to a fraction of its bandwidth
 * (MI355X box, 6.5 M-row matrix: 13 GB/s without, see bench.py).  oracle_spmv_place returns a handle for
 * oracle_spmv_placed / oracle_spmv_free. */
#include <stdlib.h>
#include <string.h>
typedef struct { int n; int* rowptr; int* col; double* val; double* x; double* y; } oracle_placed;
void* oracle_spmv_place(int n, const int* rowptr, const int* col, const double* val, const double* x) {
  oracle_placed* p = (oracle_placed*)malloc(sizeof(oracle_placed));
  const size_t nnz = (size_t)rowptr[n];
  p->n = n;
  p->rowptr = (int*)malloc(sizeof(int) * ((size_t)n + 1));
  p->col = (int*)malloc(sizeof(int) * (nnz ? nnz : 1));
  p->val = (double*)malloc(sizeof(double) * (nnz ? nnz : 1));
  p->x = (double*)malloc(sizeof(double) * (size_t)(n ? n : 1));
  p->y = (double*)malloc(sizeof(double) * (size_t)(n ? n : 1));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    p->rowptr[i] = rowptr[i];
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) { p->col[k] = col[k]; p->val[k] = val[k]; }
    p->x[i] = x[i];
    p->y[i] = 0.0;
  }
  p->rowptr[n] = rowptr[n];
  return p;
}
void oracle_spmv_placed(void* h) {
  oracle_placed* p = (oracle_placed*)h;
  oracle_csr_spmv(p->n, p->rowptr, p->col, p->val, p->x, p->y);
}
void oracle_spmv_result(void* h, double* y) { memcpy(y, ((oracle_placed*)h)->y, sizeof(double) * (size_t)((oracle_placed*)h)->n); }
void oracle_spmv_free(void* h) {
  oracle_placed* p = (oracle_placed*)h;
  free(p->rowptr); free(p->col); free(p->val); free(p->x); free(p->y); free(p);
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
