// Small dense host kernels used by the GenEO core for the projected (Rayleigh-Ritz) problems
// (<= 3m x 3m, m = LOBPCG block size) and for the replicated coarse operator E (dimE x dimE).
// Row-major storage everywhere.  These are the counterparts of the LAPACK calls PETSc/SLEPc make
// for the same tiny objects (EPS "lapack" at geneo.cpp:1193, MUMPS on E at geneo.cpp:1059-1065).
#pragma once
#include <algorithm>
#include <cmath>
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

// Symmetric eigen-decomposition A = V diag(w) V^T (cyclic Jacobi; A is n x n row-major and is
// destroyed).  Eigenvalues ascending in w, eigenvectors in the COLUMNS of v (row-major n x n).
// Jacobi is chosen for its high relative accuracy on the tiny, well-scaled projected problems.
inline void sym_eig(std::vector<double>& a, int n, std::vector<double>& w, std::vector<double>& v) {
  v.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += a[i * n + i] * a[i * n + i];
      for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    }
    if (off <= 1e-32 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (apq == 0.0) continue;
        const double app = a[p * n + p], aqq = a[q * n + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {  // rotate columns p,q of A
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {  // rotate rows p,q of A
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq;
          v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  std::vector<int> ord(n);
  for (int i = 0; i < n; ++i) ord[i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return a[x * n + x] < a[y * n + y]; });
  w.resize(n);
  std::vector<double> vs((size_t)n * n);
  for (int j = 0; j < n; ++j) {
    w[j] = a[ord[j] * n + ord[j]];
    for (int k = 0; k < n; ++k) vs[k * n + j] = v[k * n + ord[j]];
  }
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
    for (int i : cand) {
      if (used[i] && i != pick) continue;
      double s = GB[i * p + pick] * sc[i] * sc[pick];
      for (size_t k = 0; k < lrow.size(); ++k) s -= lrow[k][i] * lrow[k][pick];
      col[i] = s / piv;
    }
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
  for (int col = 0; col < r; ++col)  // M <- Lk^-1 M (forward substitution per column)
    for (int i = 0; i < r; ++i) {
      double s = M[i * r + col];
      for (int k = 0; k < i; ++k) s -= Lk[i * r + k] * M[k * r + col];
      M[i * r + col] = s / Lk[i * r + i];
    }
  for (int row = 0; row < r; ++row)  // M <- M Lk^-T
    for (int j = 0; j < r; ++j) {
      double s = M[row * r + j];
      for (int k = 0; k < j; ++k) s -= M[row * r + k] * Lk[j * r + k];
      M[row * r + j] = s / Lk[j * r + j];
    }
  for (int a = 0; a < r; ++a)  // symmetrise
    for (int b = a + 1; b < r; ++b) M[a * r + b] = M[b * r + a] = 0.5 * (M[a * r + b] + M[b * r + a]);
  std::vector<double> w, V;
  sym_eig(M, r, w, V);
  theta = w;
  // C[kept, :] = D Lk^-T V
  for (int j = 0; j < r; ++j) {
    std::vector<double> y(r);
    for (int i = 0; i < r; ++i) y[i] = V[i * r + j];
    for (int i = r - 1; i >= 0; --i) {  // solve Lk^T z = y
      double s = y[i];
      for (int k = i + 1; k < r; ++k) s -= Lk[k * r + i] * y[k];
      y[i] = s / Lk[i * r + i];
    }
    for (int i = 0; i < r; ++i) C[(size_t)kept[i] * r + j] = y[i] * sc[kept[i]];
  }
  return r;
}

}  // namespace dense
