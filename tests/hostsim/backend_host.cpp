// TEST-ONLY serial implementation of csrc/backend.h ("hostsim").
//
// Purpose: let the CPU (`-m "not gpu"`) tests exercise the HOST LOGIC of libgeneopc (core.cpp:
// setup orchestration, LOBPCG driver, batched-CG driver, E assembly, apply modes, Krylov loop,
// C-ABI argument handling) in a container without a GPU.  It is compiled only by
// tests/hostsim/build.py into tests/hostsim/libgeneopc_hostsim.so.  The product library
// (geneo4petsc_amd/csrc -> libgeneopc.so) never contains or links this file, and the
// geneo4petsc_amd package never loads the hostsim library.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "backend.h"

// GENEO_HOST_OMP (tests/hostsim/build.py build_omp -> libgeneopc_hostomp.so): the same loops with OpenMP over the rows --
// the CPU baseline bench.py times beside the GPU line (`cpu_baseline.geneo_sample`, kind "port": the library's own host
// orchestration -- LOBPCG, batched AMG-PCG, E, PCG -- on the box's host cores).  Test / bench infrastructure like the
// serial build: geneo4petsc_amd._lib.load() accepts the "hip-gfx950" backend only.  Reductions become per-thread
// partial sums combined in thread order (a different summation order than the serial build: its results agree with the
// serial ones to rounding, not to the bit).
#ifdef GENEO_HOST_OMP
#include <omp.h>
#define OMP_PRAGMA_(x) _Pragma(#x)
#define OMP_ROWS OMP_PRAGMA_(omp parallel for schedule(static))
#define OMP_SUM(v) OMP_PRAGMA_(omp parallel for schedule(static) reduction(+ : v))
#define OMP_SUM2(v, w) OMP_PRAGMA_(omp parallel for schedule(static) reduction(+ : v, w))
#else
#define OMP_ROWS
#define OMP_SUM(v)
#define OMP_SUM2(v, w)
#endif

namespace bk {

static void* g_stream = nullptr;
#ifdef GENEO_HOST_OMP
const char* name() { return "host-openmp"; }
#else
const char* name() { return "hostsim"; }
#endif
// G (pq doubles, zeroed here) += sum over rows i0..i1 of row_fn(i, accumulator): serial, or per-thread accumulators summed
// in thread order
template <class F>
static void accumulate_rows(int i0, int i1, int pq, double* g, F row_fn) {
  for (int e = 0; e < pq; ++e) g[e] = 0;
#ifdef GENEO_HOST_OMP
  const int nt = omp_get_max_threads();
  std::vector<std::vector<double>> part(nt, std::vector<double>((size_t)pq, 0.0));
#pragma omp parallel
  {
    double* my = part[omp_get_thread_num()].data();
#pragma omp for schedule(static)
    for (int i = i0; i < i1; ++i) row_fn(i, my);
  }
  for (int t = 0; t < nt; ++t)
    for (int e = 0; e < pq; ++e) g[e] += part[t][e];
#else
  for (int i = i0; i < i1; ++i) row_fn(i, g);
#endif
}
void set_stream(void* s) { g_stream = s; }
void* get_stream() { return g_stream; }
void sync() {}
// Two operators on one pattern (LOBPCG's A W / B W pass and the residual product of its lean iteration): here the shared
// pattern is a's CSR pattern and the value arrays are indexed like a.val (csr_scaled_alias sets sl_val = val).
double* sell_values_on(const Csr& a, const Csr& b) {
  if (a.n != b.n) return nullptr;
  double* out = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)a.nnz));
  for (int r = 0; r < a.n; ++r) {
    int found = 0;
    for (int k = a.rowptr[r]; k < a.rowptr[r + 1]; ++k) {
      out[k] = 0.0;
      for (int q = b.rowptr[r]; q < b.rowptr[r + 1]; ++q)
        if (b.col[q] == a.col[k]) { out[k] = b.val[q]; ++found; break; }
    }
    if (found != b.rowptr[r + 1] - b.rowptr[r]) { dfree(out); return nullptr; }   // pattern(b) not inside pattern(a)
  }
  return out;
}
bool spmm_dual_available(const Csr& a, int m) { return a.n > 0 && (m == 16 || m == 32 || m == 64); }
void spmm_dual(const Csr& a, const double* v1, const double* v2, const double* X, int ldx, double* Y1, double* Y2, int ldy,
               int m) {
  OMP_ROWS
  for (int r = 0; r < a.n; ++r)
    for (int j = 0; j < m; ++j) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = a.rowptr[r]; k < a.rowptr[r + 1]; ++k) {
        const double x = X[(int64_t)a.col[k] * ldx + j];
        s1 += v1[k] * x;
        s2 += v2[k] * x;
      }
      Y1[(int64_t)r * ldy + j] = s1;
      Y2[(int64_t)r * ldy + j] = s2;
    }
}
void spmm_dual_residual(const Csr& a, const double* v1, const double* v2, const double* X, int ldx, double* R, int ldr, int m,
                        const Chunks& c, const double* lam, const double* mask) {
  for (int s = 0; s < c.nsub; ++s) {
    OMP_ROWS
    for (int r = c.suboff[s]; r < c.suboff[s + 1]; ++r)
      for (int j = 0; j < m; ++j) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = a.rowptr[r]; k < a.rowptr[r + 1]; ++k) {
          const double x = X[(int64_t)a.col[k] * ldx + j];
          s1 += v1[k] * x;
          s2 += v2[k] * x;
        }
        R[(int64_t)r * ldr + j] = mask[s * m + j] * (s1 - lam[s * m + j] * s2);
      }
  }
}
void lobpcg_update32_basis(const Chunks& c, const double* S, const double* C, const double* keep, double* T) {
  for (int s = 0; s < c.nsub; ++s) {
    OMP_ROWS
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) {
      double pw[32];
      const double* row = S + (int64_t)i * 96;
      const double* Cs = C + (int64_t)s * 96 * 64;
      double* out = T + (int64_t)i * 96;
      for (int j = 0; j < 32; ++j) {
        double w = 0.0;
        for (int k = 32; k < 96; ++k) w += row[k] * Cs[(int64_t)k * 64 + j];
        double x = w;
        for (int k = 0; k < 32; ++k) x += row[k] * Cs[(int64_t)k * 64 + j];
        pw[j] = keep[s * 32 + j] * w;
        out[j] = x;
      }
      for (int j = 0; j < 32; ++j) out[32 + j] = pw[j];
    }
  }
}
void set_par_reduce_min(int) {}
int get_par_reduce_min() { return 0; }
int device_count() { return 0; }
int set_device(int) { return -1; }
int current_device() { return -1; }
int thread_device_check() { return -1; }
void side_stream_begin(void*, bool) {}
void side_stream_end() {}

void* alloc(size_t bytes) { return calloc(bytes ? bytes : 8, 1); }
void dfree(void* p) { free(p); }
void alloc_stats(double* a, double* f, long long* n) { *a = 0; *f = 0; *n = 0; }
void mem_info(double* live, double* live_peak, double* footprint_peak, double* cached, double* dev_free, double* dev_total, bool) {
  for (double* p : {live, live_peak, footprint_peak, cached, dev_free, dev_total})
    if (p) *p = 0.0;   // no device: the memory-bounded set-up is driven by its explicit option only
}
void h2d(void* d, const void* h, size_t b) { if (b) memcpy(d, h, b); }
void d2h(void* h, const void* d, size_t b) { if (b) memcpy(h, d, b); }
void d2d(void* d, const void* s, size_t b) { if (b) memmove(d, s, b); }
void zero(void* d, size_t b) { if (b) memset(d, 0, b); }

Csr csr_upload(int n, const int* rp, const int* col, const double* val) {
  Csr a;
  a.n = n;
  a.nnz = rp[n];
  a.rowptr = (int*)alloc(sizeof(int) * (n + 1));
  a.col = (int*)alloc(sizeof(int) * a.nnz);
  a.val = (double*)alloc(sizeof(double) * a.nnz);
  memcpy(a.rowptr, rp, sizeof(int) * (n + 1));
  memcpy(a.col, col, sizeof(int) * a.nnz);
  memcpy(a.val, val, sizeof(double) * a.nnz);
  for (int i = 0; i < n; ++i) a.max_row = std::max(a.max_row, rp[i + 1] - rp[i]);
  return a;
}
Csr csr_remap_columns(const Csr& a, const int* map) {
  Csr b = csr_upload(a.n, a.rowptr, a.col, a.val);
  for (int64_t k = 0; k < a.nnz; ++k) b.col[k] = map[a.col[k]];
  return b;
}
static Csr from_rows(int n, const std::vector<std::vector<std::pair<int, double>>>& rows) {
  std::vector<int> rp(n + 1, 0), col;
  std::vector<double> val;
  for (int i = 0; i < n; ++i) {
    for (auto& e : rows[i]) { col.push_back(e.first); val.push_back(e.second); }
    rp[i + 1] = (int)col.size();
  }
  if (col.empty()) { col.push_back(0); val.push_back(0.0); }
  return csr_upload(n, rp.data(), col.data(), val.data());
}
Csr spgemm(const Csr& a, const Csr& b, int ncols_b, bool* ok) {
  *ok = true;
  std::vector<std::vector<std::pair<int, double>>> rows(a.n);
  std::vector<int> pos(ncols_b, -1);
  for (int i = 0; i < a.n; ++i) {
    auto& r = rows[i];
    for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
      const int j = a.col[k];
      for (int l = b.rowptr[j]; l < b.rowptr[j + 1]; ++l) {
        const int c = b.col[l];
        if (pos[c] < 0) { pos[c] = (int)r.size(); r.emplace_back(c, 0.0); }
        r[pos[c]].second += a.val[k] * b.val[l];
      }
    }
    for (auto& e : r) pos[e.first] = -1;
    std::sort(r.begin(), r.end(), [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
  }
  return from_rows(a.n, rows);
}
Csr transpose(const Csr& a, int ncols, bool* ok) {
  *ok = true;
  std::vector<std::vector<std::pair<int, double>>> rows(ncols);
  for (int i = 0; i < a.n; ++i)
    for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) rows[a.col[k]].emplace_back(i, a.val[k]);
  return from_rows(ncols, rows);
}
void smooth_prolongator(Csr& ap0, const int* agg, const double* dinv, double w) {
  for (int i = 0; i < ap0.n; ++i)
    for (int k = ap0.rowptr[i]; k < ap0.rowptr[i + 1]; ++k)
      ap0.val[k] = -w * dinv[i] * ap0.val[k] + (ap0.col[k] == agg[i] ? 1.0 : 0.0);
}
void csr_finish(Csr&) {}
Csr csr_upload_raw(int n, const int* rp, const int* col, const double* val) { return csr_upload(n, rp, col, val); }
Csr csr_tentative_prolongator(int n, const int* agg) {
  std::vector<int> rp(n + 1);
  for (int i = 0; i <= n; ++i) rp[i] = i;
  std::vector<double> ones(n, 1.0);
  return csr_upload(n, rp.data(), agg, ones.data());
}
void csr_download(const Csr& a, int* rowptr, int* col, double* val) {
  memcpy(rowptr, a.rowptr, sizeof(int) * (a.n + 1));
  memcpy(col, a.col, sizeof(int) * a.nnz);
  memcpy(val, a.val, sizeof(double) * a.nnz);
}
void csr_free(Csr& a) {
  if (a.alias) { dfree(a.val); a = Csr(); return; }
  dfree(a.rowptr); dfree(a.col); dfree(a.val); a = Csr();
}
Csr csr_scaled_alias(const Csr& a, const double* rs, const double* cs, bool col_is_dinv) {
  Csr b = a;
  b.alias = true;
  b.col_scaled = col_is_dinv;
  b.val = (double*)alloc(sizeof(double) * std::max<size_t>(1, (size_t)a.nnz));
  for (int r = 0; r < a.n; ++r)
    for (int k = a.rowptr[r]; k < a.rowptr[r + 1]; ++k) b.val[k] = (rs ? rs[r] : 1.0) * a.val[k] * (cs ? cs[a.col[k]] : 1.0);
  b.sl_val = b.val;      // "values on the owner's pattern" of the two-operator products (never freed on its own)
  return b;
}
bool post_matrix(Csr& ap, const Csr& p, const double* dinv, double w) {
  for (int i = 0; i < ap.n; ++i) {
    int found = 0;
    for (int k = ap.rowptr[i]; k < ap.rowptr[i + 1]; ++k) {
      double v = -w * dinv[i] * ap.val[k];
      for (int q = p.rowptr[i]; q < p.rowptr[i + 1]; ++q)
        if (p.col[q] == ap.col[k]) { v += p.val[q]; ++found; break; }
      ap.val[k] = v;
    }
    if (found != p.rowptr[i + 1] - p.rowptr[i]) return false;
  }
  return true;
}
// no single-precision companions on the test backend: the V-cycle takes its FP64 path
bool csr_make_lp(Csr&, const Csr*) { return false; }
void csr_free_lp(Csr&) {}
void spmv_lp(const Csr&, const double*, double*) { throw std::runtime_error("hostsim: no single-precision companion"); }
void spmv_fused_lp(const Csr&, int, const double*, double*, const double*, double*, const double*, double) {
  throw std::runtime_error("hostsim: no single-precision companion");
}
void spmv(const Csr& a, const double* x, double* y) {
  OMP_ROWS
  for (int i = 0; i < a.n; ++i) {
    double s = 0;
    for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) s += a.val[k] * x[a.col[k]];
    y[i] = s;
  }
}
void spmv_profile_start(int, double) {}
void kernel_profile_start(int, double) {}
void kernel_profile_stop() {}
void kernel_profile_get(int, double* a, double* b, double* c, long long* d, long long* e) {
  if (a) *a = 0; if (b) *b = 0; if (c) *c = 0; if (d) *d = 0; if (e) *e = 0;
}
bool spmv_profiling() { return false; }
void spmv_profile_stop(double* a, double* b, long long* c, long long* d) {
  if (a) *a = 0; if (b) *b = 0; if (c) *c = 0; if (d) *d = 0;
}
void spmm_strided(const Csr& a, const double* X, int ldx, double* Y, int ldy, int m, const double* pre,
                  const double* post) {
  OMP_ROWS
  for (int i = 0; i < a.n; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        double v = a.val[k];
        if (pre) v *= pre[a.col[k]];
        s += v * X[(int64_t)a.col[k] * ldx + j];
      }
      if (post) s *= post[i];
      Y[(int64_t)i * ldy + j] = s;
    }
}
bool graph_capture_begin() { return false; }
void* graph_capture_end() { return nullptr; }
void graph_launch(void*) {}
void graph_destroy(void*) {}
bool csr_fusable(const Csr&) { return true; }
void spmm_fused(const Csr& a, int epi, const double* X, int ldx, double* Y, int ldy, int m, const double* B, int ldb,
                double* Z, int ldz, const double* dinv, double w) {
  OMP_ROWS
  for (int i = 0; i < a.n; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        const int c = a.col[k];
        s += a.val[k] * (epi == EPI_PRE ? (a.col_scaled ? 1.0 : dinv[c]) * B[(int64_t)c * ldb + j] : X[(int64_t)c * ldx + j]);
      }
      double& y = Y[(int64_t)i * ldy + j];
      if (epi == EPI_RES) y = B[(int64_t)i * ldb + j] - s;
      else if (epi == EPI_ADD) y = Z[(int64_t)i * ldz + j] + s;
      else if (epi == EPI_JAC) y = X[(int64_t)i * ldx + j] + w * dinv[i] * (B[(int64_t)i * ldb + j] - s);
      else if (epi == EPI_POST) y = w * dinv[i] * (Z[(int64_t)i * ldz + j] + B[(int64_t)i * ldb + j]) + s;
      else {
        const double bb = B[(int64_t)i * ldb + j];
        if (Z) Z[(int64_t)i * ldz + j] = w * dinv[i] * bb;
        y = bb - w * s;
      }
    }
}
void csr_diag(const Csr& a, double* d) {
  for (int i = 0; i < a.n; ++i) {
    double v = 0;
    for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k)
      if (a.col[k] == i) v += a.val[k];
    d[i] = v;
  }
}
int recip_positive(double* x, int n) {
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    if (x[i] > 0.0) x[i] = 1.0 / x[i];
    else ++bad;
  }
  return bad;
}
void gather(double* out, const double* in, const int* idx, int n) { for (int i = 0; i < n; ++i) out[i] = in[idx[i]]; }
void gather_rows(double* out, const double* in, const int* idx, int n, int w) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < w; ++j) out[(int64_t)i * w + j] = in[(int64_t)idx[i] * w + j];
}
void segsum_rows(double* out, const double* in, const int* ptr, const int* idx, int nseg, int w, bool accumulate) {
  for (int e = 0; e < nseg; ++e)
    for (int j = 0; j < w; ++j) {
      double s = 0;
      for (int k = ptr[e]; k < ptr[e + 1]; ++k) s += in[(int64_t)idx[k] * w + j];
      out[(int64_t)e * w + j] = accumulate ? out[(int64_t)e * w + j] + s : s;
    }
}
void z_rowmajor(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, double* ZR, int kp) {
  for (int s = 0; s < c.nsub; ++s) {
    const int ns = c.suboff[s + 1] - c.suboff[s];
    for (int i = 0; i < ns; ++i)
      for (int j = 0; j < kp; ++j)
        ZR[(int64_t)(c.suboff[s] + i) * kp + j] = j < ksub[s] ? Z[zbase[s] + (int64_t)j * ns + i] : 0.0;
  }
}
void gather_mul(double* out, const double* in, const int* idx, const double* d, int n) {
  for (int i = 0; i < n; ++i) out[i] = in[idx[i]] * d[i];
}
void segsum(double* out, const double* in, const int* ptr, const int* idx, int nseg, bool acc) {
  for (int e = 0; e < nseg; ++e) {
    double s = 0;
    for (int k = ptr[e]; k < ptr[e + 1]; ++k) s += in[idx[k]];
    out[e] = acc ? out[e] + s : s;
  }
}
void set(double* x, double v, int n) {
  OMP_ROWS
  for (int i = 0; i < n; ++i) x[i] = v;
}
void copy(double* y, const double* x, int n) { memmove(y, x, sizeof(double) * (size_t)n); }
void axpy(double* y, double a, const double* x, int n) {
  OMP_ROWS
  for (int i = 0; i < n; ++i) y[i] += a * x[i];
}
void axpby(double* y, double a, const double* x, double b, int n) {
  for (int i = 0; i < n; ++i) y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}
void xmy(double* y, const double* x, const double* d, int n) {
  OMP_ROWS
  for (int i = 0; i < n; ++i) y[i] = x[i] * d[i];
}
void axpy_dev(double* y, const double* a, double sign, const double* x, int n) {
  const double s = sign * a[0];
  for (int i = 0; i < n; ++i) y[i] += s * x[i];
}
void dot(const double* x, const double* y, int n, double* out) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  out[0] = s;
}

Chunks chunks_upload(int nsub, const int* off) {
  Chunks c;
  c.nsub = nsub;
  c.n = off[nsub] - off[0];
  std::vector<int> st, ln, sb, sp;
  sp.push_back(0);
  for (int s = 0; s < nsub; ++s) {
    for (int a = off[s]; a < off[s + 1]; a += CHUNK) {
      st.push_back(a);
      ln.push_back(std::min(CHUNK, off[s + 1] - a));
      sb.push_back(s);
    }
    sp.push_back((int)st.size());
  }
  c.nchunk = (int)st.size();
  auto dup = [](const std::vector<int>& v) {
    int* p = (int*)alloc(sizeof(int) * std::max<size_t>(1, v.size()));
    if (!v.empty()) memcpy(p, v.data(), sizeof(int) * v.size());
    return p;
  };
  c.start = dup(st); c.len = dup(ln); c.sub = dup(sb); c.subptr = dup(sp);
  c.suboff = (int*)alloc(sizeof(int) * (nsub + 1));
  memcpy(c.suboff, off, sizeof(int) * (nsub + 1));
  c.partial = (double*)alloc(sizeof(double) * 4 * std::max(std::max(1, c.nchunk), nsub));
  return c;
}
void chunks_free(Chunks& c) {
  dfree(c.start); dfree(c.len); dfree(c.sub); dfree(c.subptr); dfree(c.suboff); dfree(c.partial);
  c = Chunks();
}
void seg_dot(const Chunks& c, const double* x, const double* y, double* out, int stride, int slot) {
  for (int s = 0; s < c.nsub; ++s) {
    double t = 0;
    OMP_SUM(t)
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) t += x[i] * y[i];
    out[(int64_t)s * stride + slot] = t;
  }
}
void cg_start(const Chunks& c, double* sc, double* x, double* r, double* z, double* p, const double* b,
              const double* dinv) {
  for (int s = 0; s < c.nsub; ++s) {
    double rz = 0, rr = 0;
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) {
      x[i] = 0; r[i] = b[i];
      if (dinv) { z[i] = dinv[i] * r[i]; p[i] = z[i]; rz += r[i] * z[i]; }
      rr += r[i] * r[i];
    }
    double* q = sc + (int64_t)s * 8;
    q[0] = q[1] = rz; q[2] = 0; q[3] = rr; q[4] = q[5] = 0; q[6] = rr > 0 ? 1.0 : 0.0; q[7] = rr;
  }
}
void seg_partial(const Chunks& c, const double* x, const double* y, int slot) {
  for (int s = 0; s < c.nsub; ++s) {
    double t = 0;
    OMP_SUM(t)
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) t += x[i] * y[i];
    if (slot == 0) c.partial[s] = t;
    else c.partial[c.nsub + 2 * s + (slot - 1)] = t;
  }
}
void cg_set_rz(const Chunks& c, double* sc) {
  for (int s = 0; s < c.nsub; ++s) sc[(int64_t)s * 8] = sc[(int64_t)s * 8 + 1] = c.partial[c.nsub + 2 * s];
}
void dense_sym_apply(const Chunks& c, const double* inv, const int64_t* base, const double* B, int ldb, double* X,
                     int ldx, int m) {
  for (int s = 0; s < c.nsub; ++s) {
    const int s0 = c.suboff[s], ns = c.suboff[s + 1] - s0;
    const double* A = inv + base[s];
    for (int i = 0; i < ns; ++i)
      for (int j = 0; j < m; ++j) {
        double acc = 0;
        for (int k = 0; k < ns; ++k) acc += A[(int64_t)k * ns + i] * B[(int64_t)(s0 + k) * ldb + j];
        X[(int64_t)(s0 + i) * ldx + j] = acc;
      }
  }
}
void seg_pap(const Chunks& c, const double* p, const double* q) {
  for (int s = 0; s < c.nsub; ++s) {
    double t = 0;
    OMP_SUM(t)
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) t += p[i] * q[i];
    c.partial[s] = t;  // hostsim keeps one partial per subdomain
  }
}
void cg_update(const Chunks& c, double* sc, int parity, double* x, double* r, double* z, const double* p,
               const double* q, const double* dinv) {
  for (int s = 0; s < c.nsub; ++s) {
    double* t = sc + (int64_t)s * 8;
    const double pap = c.partial[s];
    const double alpha = (t[6] != 0.0 && pap != 0.0) ? t[parity] / pap : 0.0;
    double nrz = 0, nrr = 0;
    OMP_SUM2(nrz, nrr)
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) {
      x[i] += alpha * p[i];
      r[i] -= alpha * q[i];
      if (dinv) { z[i] = dinv[i] * r[i]; nrz += r[i] * z[i]; }
      nrr += r[i] * r[i];
    }
    t[2] = pap; t[4] = alpha;
    if (dinv) c.partial[c.nsub + 2 * s] = nrz;
    c.partial[c.nsub + 2 * s + 1] = nrr;
  }
}
void cg_direction(const Chunks& c, double* sc, int parity, double* p, const double* z, double tol2) {
  for (int s = 0; s < c.nsub; ++s) {
    double* t = sc + (int64_t)s * 8;
    const double nrz = c.partial[c.nsub + 2 * s], nrr = c.partial[c.nsub + 2 * s + 1];
    const double rz = t[parity];
    const double beta = (t[6] != 0.0 && rz != 0.0) ? nrz / rz : 0.0;
    if (t[6] != 0.0) {
      OMP_ROWS
      for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) p[i] = z[i] + beta * p[i];
    }
    t[parity ^ 1] = nrz; t[3] = nrr; t[5] = beta;
    if (t[6] != 0.0 && nrr <= tol2 * t[7]) t[6] = 0.0;
  }
}

void gram(const Chunks& c, const double* S, int lds, int p, const double* T, int ldt, int q, double* G) {
  for (int s = 0; s < c.nsub; ++s) {
    accumulate_rows(c.suboff[s], c.suboff[s + 1], p * q, G + (int64_t)s * p * q, [&](int i, double* g) {
      for (int a = 0; a < p; ++a) {
        const double sv = S[(int64_t)i * lds + a];
        if (sv == 0.0) continue;
        for (int b = 0; b < q; ++b) g[a * q + b] += sv * T[(int64_t)i * ldt + b];
      }
    });
  }
}
void gram2(const Chunks& c, const double* S1, int lds1, int p1, const double* S2, int lds2, int p2, const double* T, int ldt,
           int q, double* G) {
  const int p = p1 + p2;
  for (int s = 0; s < c.nsub; ++s) {
    accumulate_rows(c.suboff[s], c.suboff[s + 1], p * q, G + (int64_t)s * p * q, [&](int i, double* g) {
      for (int a = 0; a < p; ++a) {
        const double sv = a < p1 ? S1[(int64_t)i * lds1 + a] : S2[(int64_t)i * lds2 + a - p1];
        if (sv == 0.0) continue;
        for (int b = 0; b < q; ++b) g[a * q + b] += sv * T[(int64_t)i * ldt + b];
      }
    });
  }
}
void block_mul(const Chunks& c, const double* S, int lds, int p, const double* C, int q, double* Y, int ldy,
               bool acc) {
  for (int s = 0; s < c.nsub; ++s) {
    const double* cs = C + (int64_t)s * p * q;
    OMP_ROWS
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) {
      std::vector<double> row(q);
      for (int j = 0; j < q; ++j) row[j] = acc ? Y[(int64_t)i * ldy + j] : 0.0;
      for (int k = 0; k < p; ++k) {
        const double sv = S[(int64_t)i * lds + k];
        if (sv == 0.0) continue;
        for (int j = 0; j < q; ++j) row[j] += sv * cs[k * q + j];
      }
      for (int j = 0; j < q; ++j) Y[(int64_t)i * ldy + j] = row[j];
    }
  }
}
void block_residual(const Chunks& c, const double* AX, int lda, const double* BX, int ldb, const double* lam, int m,
                    double* R, int ldr, double* nrm) {
  for (int s = 0; s < c.nsub; ++s)
    OMP_ROWS
    for (int j = 0; j < m; ++j) {
      double t = 0;
      const double lj = lam[(int64_t)s * m + j];
      for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) {
        const double v = AX[(int64_t)i * lda + j] - lj * BX[(int64_t)i * ldb + j];
        R[(int64_t)i * ldr + j] = v;
        t += v * v;
      }
      nrm[(int64_t)s * m + j] = t;
    }
}
void block_residual_norms(const Chunks& c, const double* AX, int lda, const double* BX, int ldb, const double* lam,
                          int m, double* R, int ldr, const double* colmask, double* nrm3) {
  for (int s = 0; s < c.nsub; ++s)
    OMP_ROWS
    for (int j = 0; j < m; ++j) {
      double tr = 0, ta = 0, tb = 0;
      const double lj = lam[(int64_t)s * m + j];
      const double mk = colmask ? colmask[(int64_t)s * m + j] : 1.0;
      for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) {
        const double a = AX[(int64_t)i * lda + j], b = BX[(int64_t)i * ldb + j];
        const double v = a - lj * b;
        R[(int64_t)i * ldr + j] = mk * v;
        tr += v * v; ta += a * a; tb += b * b;
      }
      if (nrm3) {
        nrm3[(int64_t)s * 3 * m + j] = tr;
        nrm3[(int64_t)s * 3 * m + m + j] = ta;
        nrm3[(int64_t)s * 3 * m + 2 * m + j] = tb;
      }
    }
}
void block_colnorm(const Chunks& c, const double* X, int ldx, int m, double* nrm) {
  for (int s = 0; s < c.nsub; ++s)
    for (int j = 0; j < m; ++j) {
      double t = 0;
      for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i) t += X[(int64_t)i * ldx + j] * X[(int64_t)i * ldx + j];
      nrm[(int64_t)s * m + j] = t;
    }
}
void block_axpby(double* Y, int ldy, double a, const double* X, int ldx, double b, int n, int m) {
  OMP_ROWS
  for (int64_t i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const double xv = a * X[i * ldx + j];
      Y[i * ldy + j] = (b == 0.0) ? xv : xv + b * Y[i * ldy + j];
    }
}
void block_rowscale(double* Y, int ldy, const double* X, int ldx, const double* d, double a, double b, int n,
                    int m) {
  OMP_ROWS
  for (int64_t i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const double xv = a * d[i] * X[i * ldx + j];
      Y[i * ldy + j] = (b == 0.0) ? xv : xv + b * Y[i * ldy + j];
    }
}
void jacobi_step(double* X, int ldx, const double* B, int ldb, const double* AX, const double* dinv, double w, int n,
                 int m, bool zero_guess) {
  OMP_ROWS
  for (int64_t i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const double bv = B[i * ldb + j];
      if (zero_guess) X[i * ldx + j] = w * dinv[i] * bv;
      else X[i * ldx + j] += w * dinv[i] * (bv - AX[i * m + j]);
    }
}
void cheb_update(double* r, const double* ad, double* d, double* z, int ldz, const double* dinv, double a, double b,
                 int n, int m) {
  OMP_ROWS
  for (int64_t i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const int64_t e = i * m + j;
      r[e] -= ad[e];
      d[e] = a * dinv[i] * r[e] + b * d[e];
      z[i * ldz + j] += d[e];
    }
}
void block_colscale(const Chunks& c, double* X, int ldx, int m, const double* cs) {
  for (int s = 0; s < c.nsub; ++s)
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i)
      for (int j = 0; j < m; ++j) X[(int64_t)i * ldx + j] *= cs[(int64_t)s * m + j];
}
static inline double hash_unit(uint64_t seed, uint64_t gid, uint64_t row, uint64_t colj) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (gid + 1) + 0xBF58476D1CE4E5B9ull * (row + 1) +
               0x94D049BB133111EBull * (colj + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}
double hash_unit_host(uint64_t seed, uint64_t gid, uint64_t row, uint64_t colj) {
  return hash_unit(seed, gid, row, colj);
}
void block_init(const Chunks& c, double* X, int ldx, int m, const int* sub_gid, uint64_t seed) {
  for (int s = 0; s < c.nsub; ++s)
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i)
      for (int j = 0; j < m; ++j)
        X[(int64_t)i * ldx + j] =
            (j == 0) ? 1.0 : hash_unit(seed, (uint64_t)sub_gid[s], (uint64_t)(i - c.suboff[s]), (uint64_t)j);
}
void block_extract(const Chunks& c, const double* X, int ldx, int m, const double* d, const int* sel,
                   const int* ksub, const int64_t* zbase, double* Z) {
  for (int s = 0; s < c.nsub; ++s) {
    const int ns = c.suboff[s + 1] - c.suboff[s];
    for (int j = 0; j < ksub[s]; ++j) {
      const int src = sel[(int64_t)s * m + j];
      for (int i = 0; i < ns; ++i) {
        const int64_t row = c.suboff[s] + i;
        Z[zbase[s] + (int64_t)j * ns + i] = d[row] * (src < 0 ? 1.0 : X[row * ldx + src]);
      }
    }
  }
}
void zt_apply(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, const int* zoff, int kmax,
              const double* xL, double* yE, int dimE_total) {
  (void)kmax;
  for (int e = 0; e < dimE_total; ++e) yE[e] = 0;
  for (int s = 0; s < c.nsub; ++s) {
    const int ns = c.suboff[s + 1] - c.suboff[s];
    for (int j = 0; j < ksub[s]; ++j) {
      double t = 0;
      for (int i = 0; i < ns; ++i) t += Z[zbase[s] + (int64_t)j * ns + i] * xL[c.suboff[s] + i];
      yE[zoff[s] + j] = t;
    }
  }
}
void z_apply(const Chunks& c, const double* Z, const int64_t* zbase, const int* ksub, const int* zoff,
             const double* yE, double* wL, bool acc) {
  for (int s = 0; s < c.nsub; ++s) {
    const int ns = c.suboff[s + 1] - c.suboff[s];
    for (int i = 0; i < ns; ++i) {
      double a = 0;
      for (int j = 0; j < ksub[s]; ++j) a += Z[zbase[s] + (int64_t)j * ns + i] * yE[zoff[s] + j];
      wL[c.suboff[s] + i] = acc ? wL[c.suboff[s] + i] + a : a;
    }
  }
}
void set_mfma(bool) {}
bool set_variant(const char*, int) { return false; }
bool chol_solve(const double* L, const double* LT, int n, double* y) {   // the loops of dense::cholesky_solve_lu
  (void)LT;
  for (int i = 0; i < n; ++i) {
    const double* li = L + (size_t)i * n;
    double s = y[i];
    for (int k = 0; k < i; ++k) s -= li[k] * y[k];
    y[i] = s / li[i];
  }
  for (int k = n - 1; k >= 0; --k) {
    const double* lk = L + (size_t)k * n;
    const double xk = y[k] / lk[k];
    y[k] = xk;
    for (int i = 0; i < k; ++i) y[i] -= lk[i] * xk;
  }
  return true;
}
void* pinned_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 8); }
void pinned_free(void* p) { std::free(p); }
void h2d_async(void* d, const void* h, size_t bytes) { if (bytes) std::memcpy(d, h, bytes); }
void set_spmv_kind(int) {}
const char* spmv_kernel_name() { return "hostsim"; }
int selftest_mfma_f64() { return 0; }
void* event_create() { return nullptr; }
void event_record(void*) {}
void alloc_cache_release() {}
void d2h_after(void* h, const void* d, size_t b, void*) { if (b) memcpy(h, d, b); }
float event_elapsed_ms(void*, void*) { return 0.f; }
void event_destroy(void*) {}
bool lobpcg_update32_available() { return true; }
void lobpcg_update32(const Chunks& c, const double* S, const double* AS, const double* BS, const double* C,
                     const double* keep, const double* lam, const double* mask, double* T, double* AT, double* BT,
                     double* R) {
  const double* src[3] = {S, AS, BS};
  double* dst[3] = {T, AT, BT};
  std::vector<double> ax(32), pw(32);
  for (int s = 0; s < c.nsub; ++s)
    for (int i = c.suboff[s]; i < c.suboff[s + 1]; ++i)
      for (int op = 0; op < 3; ++op) {
        const double* row = src[op] + (int64_t)i * 96;
        const double* Cs = C + (int64_t)s * 96 * 64;
        double* out = dst[op] + (int64_t)i * 96;
        for (int j = 0; j < 32; ++j) {
          double w = 0.0;
          for (int k = 32; k < 96; ++k) w += row[k] * Cs[(int64_t)k * 64 + j];
          double x = w;
          for (int k = 0; k < 32; ++k) x += row[k] * Cs[(int64_t)k * 64 + j];
          pw[j] = keep[s * 32 + j] * w;
          out[j] = x;
          if (op == 1) ax[j] = x;
          if (op == 2) R[(int64_t)i * 32 + j] = (mask ? mask[s * 32 + j] : 1.0) * (ax[j] - lam[s * 32 + j] * x);
        }
        for (int j = 0; j < 32; ++j) out[32 + j] = pw[j];
      }
}

}  // namespace bk
