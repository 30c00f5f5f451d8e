// Small dense host kernels used by the GenEO core for the projected (Rayleigh-Ritz) problems
// (<= 3m x 3m, m = LOBPCG block size) and for the replicated coarse operator E (dimE x dimE).
// Row-major storage everywhere.  These are the counterparts of the LAPACK calls PETSc/SLEPc make
// for the same tiny objects (EPS "lapack" at geneo.cpp:1193, MUMPS on E at geneo.cpp:1059-1065).
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <thread>
#include <vector>

namespace dense {

// Cholesky A = L L^T (lower, in place, upper part untouched).  Returns false if not SPD.
inline bool cholesky(std::vector<double>& a, int n) {
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    a[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[i * n + j];
      for (int k = 0; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = s / d;
    }
  }
  return true;
}
// Cholesky with NULL-PIVOT FIXING, the counterpart of what the reference asks of MUMPS for its local factorisations
// (tuneSolver, geneo.cpp:76-92: ICNTL(24) = 1 detects null pivots, CNTL(5) = 1e20 replaces them by a huge value): a
// pivot below tol * max|diag| is replaced by 1e20 * max|diag| and its column is not eliminated, so the matching unknown
// comes out ~0 and the factor represents a bounded generalised inverse of a singular (e.g. pure Neumann) block.  On a
// definite matrix no pivot is null and the result is bit-identical to cholesky().  Returns the number of fixed pivots,
// -1 when a pivot is NEGATIVE beyond the threshold (indefinite matrix).
inline int cholesky_fix_null_pivots(std::vector<double>& a, int n, double tol = 1e-12) {
  double dmax = 0.0;
  for (int j = 0; j < n; ++j) dmax = std::max(dmax, std::fabs(a[j * n + j]));
  const double thr = tol * dmax;
  int fixed = 0;
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    if (d < -thr) return -1;
    if (!(d > thr)) {                     // null pivot: pinned
      ++fixed;
      a[j * n + j] = 1e10 * std::sqrt(dmax > 0.0 ? dmax : 1.0);
      for (int i = j + 1; i < n; ++i) a[i * n + j] = 0.0;
      continue;
    }
    d = std::sqrt(d);
    a[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[i * n + j];
      for (int k = 0; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = s / d;
    }
  }
  return fixed;
}
// solve L L^T x = b in place
inline void cholesky_solve(const std::vector<double>& l, int n, double* x) {
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int k = 0; k < i; ++k) s -= l[i * n + k] * x[k];
    x[i] = s / l[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= l[k * n + i] * x[k];
    x[i] = s / l[i * n + i];
  }
}

// Blocked right-looking Cholesky for the replicated coarse operator E when it gets large (dimE = 20 per subdomain:
// 1280 on 8 GPUs): the panel solve and the trailing update run over row ranges on host threads, every inner loop is
// a contiguous dot product of two 64-long row pieces.  Same result layout as cholesky() (lower, in place).
inline bool cholesky_blocked(std::vector<double>& a, int n, int nthreads) {
  constexpr int NB = 64;
  if (n < 4 * NB || nthreads <= 1) return cholesky(a, n);
  auto par = [&](int lo, int hi, const std::function<void(int)>& f) {   // rows lo..hi-1 dealt round-robin (triangular work)
    const int nt = std::max(1, std::min(nthreads, hi - lo));
    if (nt == 1) {
      for (int i = lo; i < hi; ++i) f(i);
      return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t]() {
        for (int i = lo + t; i < hi; i += nt) f(i);
      });
    for (auto& x : th) x.join();
  };
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0), k1 = k0 + kb;
    for (int j = k0; j < k1; ++j) {                      // diagonal block, unblocked
      double d = a[(size_t)j * n + j];
      for (int k = k0; k < j; ++k) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
      if (!(d > 0.0)) return false;
      d = std::sqrt(d);
      a[(size_t)j * n + j] = d;
      for (int i = j + 1; i < k1; ++i) {
        double s2 = a[(size_t)i * n + j];
        for (int k = k0; k < j; ++k) s2 -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
        a[(size_t)i * n + j] = s2 / d;
      }
    }
    if (k1 >= n) break;
    par(k1, n, [&](int i) {                               // panel: L[i, k0:k1] = A[i, k0:k1] L_kk^-T
      double* ri = a.data() + (size_t)i * n;
      for (int c = k0; c < k1; ++c) {
        const double* rc = a.data() + (size_t)c * n;
        double s2 = ri[c];
        for (int t = k0; t < c; ++t) s2 -= ri[t] * rc[t];
        ri[c] = s2 / rc[c];
      }
    });
    par(k1, n, [&](int i) {                               // trailing update, lower triangle: A[i, j] -= L[i,blk] . L[j,blk]
      double* ri = a.data() + (size_t)i * n;
      int j = k1;
      for (; j + 3 <= i; j += 4) {                        // four rows of L per pass: the 64-long piece of row i is reused
        const double* r0 = a.data() + (size_t)j * n;
        const double *r1 = r0 + n, *r2 = r1 + n, *r3 = r2 + n;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int t = k0; t < k1; ++t) {
          const double v = ri[t];
          s0 += v * r0[t];
          s1 += v * r1[t];
          s2 += v * r2[t];
          s3 += v * r3[t];
        }
        ri[j] -= s0; ri[j + 1] -= s1; ri[j + 2] -= s2; ri[j + 3] -= s3;
      }
      for (; j <= i; ++j) {
        const double* rj = a.data() + (size_t)j * n;
        double s2 = 0.0;
        for (int t = k0; t < k1; ++t) s2 += ri[t] * rj[t];
        ri[j] -= s2;
      }
    });
  }
  return true;
}
// solve L L^T x = b with U = L^T stored row-major next to L: both substitutions walk contiguous rows
inline void cholesky_solve_lu(const std::vector<double>& l, const std::vector<double>& u, int n, double* x) {
  for (int i = 0; i < n; ++i) {
    const double* li = l.data() + (size_t)i * n;
    double s = x[i];
    for (int k = 0; k < i; ++k) s -= li[k] * x[k];
    x[i] = s / li[i];
  }
  // backward sweep column by column, from the last unknown down (x_i -= L_ki x_k for k = n - 1, n - 2, ..): the order
  // the device kernel of the same solve uses (bk::chol_solve), so that both produce the same bits; column k of L^T is
  // row k of L, contiguous
  (void)u;
  for (int k = n - 1; k >= 0; --k) {
    const double* lk = l.data() + (size_t)k * n;
    const double xk = x[k] / lk[k];
    x[k] = xk;
    for (int i = 0; i < k; ++i) x[i] -= lk[i] * xk;
  }
}

// LU with partial pivoting (fallback for E when Cholesky fails).  Returns false if singular.
inline bool lu_factor(std::vector<double>& a, int n, std::vector<int>& piv) {
  piv.resize(n);
  for (int k = 0; k < n; ++k) {
    int p = k;
    double mx = std::fabs(a[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[i * n + k]) > mx) { mx = std::fabs(a[i * n + k]); p = i; }
    piv[k] = p;
    if (mx == 0.0) return false;
    if (p != k)
      for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
    for (int i = k + 1; i < n; ++i) {
      const double f = a[i * n + k] / a[k * n + k];
      a[i * n + k] = f;
      if (f != 0.0)
        for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
    }
  }
  return true;
}
inline void lu_solve(const std::vector<double>& a, int n, const std::vector<int>& piv, double* x) {
  for (int k = 0; k < n; ++k)
    if (piv[k] != k) std::swap(x[k], x[piv[k]]);
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int k = 0; k < i; ++k) s -= a[i * n + k] * x[k];
    x[i] = s;
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= a[i * n + k] * x[k];
    x[i] = s / a[i * n + i];
  }
}

// sum a[k] b[k] with eight independent partial sums in a fixed order (vectorisable without re-association flags)
inline double dot8(const double* a, const double* b, int n) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int k = 0;
  for (; k + 8 <= n; k += 8)
    for (int u = 0; u < 8; ++u) acc[u] += a[k + u] * b[k + u];
  double s = 0.0;
  for (; k < n; ++k) s += a[k] * b[k];
  return s + ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
}

// Symmetric eigen-decomposition A = V diag(w) V^T: Householder tridiagonalisation followed by the
// implicit QL iteration (the classical tred2 / tql2 pair).  A is n x n row-major and is destroyed.
// Eigenvalues ascending in w, eigenvectors in the COLUMNS of v (row-major n x n).
inline void sym_eig(std::vector<double>& a, int n, std::vector<double>& w, std::vector<double>& v) {
  // The algorithm walks down columns (first index) in its inner loops: the working copy is held
  // transposed (A is symmetric, so the input needs no transposition) to make those walks contiguous
  // and vectorisable; the eigenvectors are transposed back while sorting.
  v = a;
  w.assign(n, 0.0);
  std::vector<double> e(n, 0.0), rc(n, 0.0), rs(n, 0.0);
  auto V = [&](int i, int j) -> double& { return v[(size_t)j * n + i]; };
  if (n == 0) return;
  // ---- tred2
  for (int j = 0; j < n; ++j) w[j] = V(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(w[k]);
    if (scale == 0.0) {
      e[i] = w[i - 1];
      for (int j = 0; j < i; ++j) {
        w[j] = V(i - 1, j);
        V(i, j) = 0.0;
        V(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; ++k) {
        w[k] /= scale;
        h += w[k] * w[k];
      }
      double f = w[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      w[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = w[j];
        V(j, i) = f;
        const double* colj = v.data() + (size_t)j * n;      // V(k, j), contiguous in k
        g = e[j] + colj[j] * f + dot8(colj + j + 1, w.data() + j + 1, i - 1 - j);
        for (int k = j + 1; k <= i - 1; ++k) e[k] += colj[k] * f;
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * w[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * w[j];
      for (int j = 0; j < i; ++j) {
        f = w[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) V(k, j) -= (f * e[k] + g * w[k]);
        w[j] = V(i - 1, j);
        V(i, j) = 0.0;
      }
    }
    w[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    V(n - 1, i) = V(i, i);
    V(i, i) = 1.0;
    const double h = w[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) w[k] = V(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double* colj = v.data() + (size_t)j * n;
        const double g = dot8(v.data() + (size_t)(i + 1) * n, colj, i + 1);
        for (int k = 0; k <= i; ++k) colj[k] -= g * w[k];
      }
    }
    for (int k = 0; k <= i; ++k) V(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) {
    w[j] = V(n - 1, j);
    V(n - 1, j) = 0.0;
  }
  V(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- tql2
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(w[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m >= n) m = n - 1;
    if (m > l) {
      int iter = 0;
      do {
        ++iter;
        double g = w[l];
        double p = (w[l + 1] - g) / (2.0 * e[l]);
        double r = std::sqrt(p * p + 1.0);
        if (p < 0) r = -r;
        w[l] = e[l] / (p + r);
        w[l + 1] = e[l] * (p + r);
        const double dl1 = w[l + 1];
        double h = g - w[l];
        for (int i = l + 2; i < n; ++i) w[i] -= h;
        f += h;
        p = w[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        // The scalar recurrence of the sweep first (its chain of square roots and divisions is the critical path), the
        // plane rotations of the eigenvector columns afterwards, a strip of rows at a time: the strip of column i + 1
        // stays in registers while the sweep walks down the columns -- one load and one store per column and strip
        // instead of two each, and the vector work no longer waits for the scalar chain.
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::sqrt(p * p + e[i] * e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * w[i] - s * g;
          w[i + 1] = h + s * (c * g + s * w[i]);
          rc[i] = c;
          rs[i] = s;
        }
        constexpr int STRIP = 16;
        for (int k0 = 0; k0 < n; k0 += STRIP) {
          const int kw = std::min(STRIP, n - k0);
          double hh[STRIP];
          const double* top = v.data() + (size_t)m * n + k0;
          for (int k = 0; k < kw; ++k) hh[k] = top[k];
          if (kw == STRIP) {
            for (int i = m - 1; i >= l; --i) {
              double* vi = v.data() + (size_t)i * n + k0;
              double* vi1 = vi + n;
              const double ci = rc[i], si = rs[i];
              for (int k = 0; k < STRIP; ++k) {
                const double x = vi[k];
                vi1[k] = si * x + ci * hh[k];
                hh[k] = ci * x - si * hh[k];
              }
            }
          } else {
            for (int i = m - 1; i >= l; --i) {
              double* vi = v.data() + (size_t)i * n + k0;
              double* vi1 = vi + n;
              const double ci = rc[i], si = rs[i];
              for (int k = 0; k < kw; ++k) {
                const double x = vi[k];
                vi1[k] = si * x + ci * hh[k];
                hh[k] = ci * x - si * hh[k];
              }
            }
          }
          double* bot = v.data() + (size_t)l * n + k0;
          for (int k = 0; k < kw; ++k) bot[k] = hh[k];
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        w[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
    }
    w[l] += f;
    e[l] = 0.0;
  }
  // sort ascending
  std::vector<int> ord(n);
  for (int i = 0; i < n; ++i) ord[i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return w[x] < w[y]; });
  std::vector<double> ws(n), vs((size_t)n * n);
  for (int j = 0; j < n; ++j) {
    ws[j] = w[ord[j]];
    const double* src = v.data() + (size_t)ord[j] * n;   // eigenvector ord[j], contiguous in the working copy
    for (int k = 0; k < n; ++k) vs[(size_t)k * n + j] = src[k];
  }
  w.swap(ws);
  v.swap(vs);
}

// Rank-revealing generalized symmetric eigenproblem  GA c = theta GB c  on a p x p pencil whose
// GB may be numerically rank-deficient (converged / dependent LOBPCG basis columns).
//  1. D = diag(GB)^-1/2 scaling (zero-norm columns dropped)
//  2. Cholesky of the scaled GB with diagonal pivoting restricted to columns >= nfix (the first
//     nfix columns -- the current Ritz vectors -- are taken in order); columns whose pivot falls
//     below `drop` are discarded
//  3. standard eigenproblem L^-1 GA L^-T y = theta y on the kept columns
// Output: theta ascending (r values), C (p x r, row-major): S*C are GB-orthonormal Ritz vectors.
inline int gen_eig_rr(const std::vector<double>& GA, const std::vector<double>& GB, int p, int nfix, double drop,
                      std::vector<double>& theta, std::vector<double>& C) {
  std::vector<double> sc(p, 0.0);
  std::vector<int> cand;
  for (int i = 0; i < p; ++i) {
    const double d = GB[i * p + i];
    if (d > 0.0 && std::isfinite(d)) {
      sc[i] = 1.0 / std::sqrt(d);
      cand.push_back(i);
    }
  }
  // pivoted Cholesky on the scaled matrix, working copy indexed by original ids
  const int nc = (int)cand.size();
  std::vector<int> kept;
  std::vector<double> L((size_t)nc * nc, 0.0);  // rows: kept order
  std::vector<double> dg(p, 0.0);
  for (int i : cand) dg[i] = 1.0;  // scaled diagonal
  std::vector<char> used(p, 0);
  std::vector<std::vector<double>> lrow;  // lrow[k][i] = L(i, k) for original id i (column k of L)
  for (int step = 0; step < nc; ++step) {
    int pick = -1;
    // fixed-order prefix first
    for (int i : cand)
      if (!used[i] && i < nfix) { pick = i; break; }
    if (pick < 0) {
      double best = -1.0;
      for (int i : cand)
        if (!used[i] && dg[i] > best) { best = dg[i]; pick = i; }
    }
    if (pick < 0) break;
    used[pick] = 1;
    if (!(dg[pick] > drop)) {
      if (pick < nfix) continue;  // a dependent Ritz vector: skip it, keep going
      break;                      // pivoted part: everything that remains is smaller
    }
    const double piv = std::sqrt(dg[pick]);
    std::vector<double> col(p, 0.0);
    for (int i : cand) col[i] = GB[i * p + pick] * sc[i] * sc[pick];
    for (size_t k = 0; k < lrow.size(); ++k) {       // col -= L(:, k) L(pick, k), contiguous in i
      const double f = lrow[k][pick];
      const double* lk = lrow[k].data();
      for (int i = 0; i < p; ++i) col[i] -= lk[i] * f;
    }
    for (int i : cand) col[i] = (used[i] && i != pick) ? 0.0 : col[i] / piv;
    for (int i = 0; i < p; ++i)
      if (sc[i] == 0.0) col[i] = 0.0;
    col[pick] = piv;
    for (int i : cand)
      if (!used[i]) dg[i] -= col[i] * col[i];
    lrow.push_back(col);
    kept.push_back(pick);
  }
  const int r = (int)kept.size();
  theta.clear();
  C.assign((size_t)p * std::max(r, 1), 0.0);
  if (r == 0) return 0;
  // Lk (r x r lower) with Lk(a,b) = lrow[b][kept[a]]
  std::vector<double> Lk((size_t)r * r, 0.0);
  for (int a = 0; a < r; ++a)
    for (int b = 0; b <= a; ++b) Lk[a * r + b] = lrow[b][kept[a]];
  // M = Lk^-1 (D GA D)[kept,kept] Lk^-T
  std::vector<double> M((size_t)r * r);
  for (int a = 0; a < r; ++a)
    for (int b = 0; b < r; ++b) M[a * r + b] = GA[kept[a] * p + kept[b]] * sc[kept[a]] * sc[kept[b]];
  auto lower_solve_rows = [&](std::vector<double>& X) {  // X <- Lk^-1 X, whole rows at a time (contiguous)
    for (int i = 0; i < r; ++i) {
      double* xi = X.data() + (size_t)i * r;
      for (int k = 0; k < i; ++k) {
        const double f = Lk[i * r + k];
        const double* xk = X.data() + (size_t)k * r;
        for (int c = 0; c < r; ++c) xi[c] -= f * xk[c];
      }
      const double inv = 1.0 / Lk[i * r + i];
      for (int c = 0; c < r; ++c) xi[c] *= inv;
    }
  };
  lower_solve_rows(M);                                   // T = Lk^-1 M
  {                                                      // Lk^-1 M Lk^-T = Lk^-1 T^T (the result is symmetric)
    std::vector<double> Tt((size_t)r * r);
    for (int a = 0; a < r; ++a)
      for (int b = 0; b < r; ++b) Tt[(size_t)b * r + a] = M[(size_t)a * r + b];
    M.swap(Tt);
  }
  lower_solve_rows(M);
  for (int a = 0; a < r; ++a)  // symmetrise
    for (int b = a + 1; b < r; ++b) M[a * r + b] = M[b * r + a] = 0.5 * (M[a * r + b] + M[b * r + a]);
  std::vector<double> w, V;
  sym_eig(M, r, w, V);
  theta = w;
  // C[kept, :] = D Lk^-T V : back substitution on all eigenvectors at once, row by row
  for (int i = r - 1; i >= 0; --i) {
    double* zi = V.data() + (size_t)i * r;
    for (int k = i + 1; k < r; ++k) {
      const double f = Lk[k * r + i];
      const double* zk = V.data() + (size_t)k * r;
      for (int c = 0; c < r; ++c) zi[c] -= f * zk[c];
    }
    const double inv = 1.0 / Lk[i * r + i];
    for (int c = 0; c < r; ++c) zi[c] *= inv;
  }
  for (int i = 0; i < r; ++i) {
    const double f = sc[kept[i]];
    for (int j = 0; j < r; ++j) C[(size_t)kept[i] * r + j] = V[(size_t)i * r + j] * f;
  }
  return r;
}

}  // namespace dense
