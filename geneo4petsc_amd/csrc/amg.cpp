// Smoothed-aggregation AMG: host set-up + device V-cycle (see amg.h).
#include "amg.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "dense.h"

namespace geneo {

static std::atomic<int> g_null_pivots{0};     // null pivots fixed in the coarsest blocks since the last amg_null_pivots_take()
int amg_null_pivots_take() { return g_null_pivots.exchange(0); }

// ------------------------------------------------------------------------------ host sparse kernels
// f(t, r0, r1) over contiguous row ranges on up to 16 host threads (serial below 20000 rows)
static void parallel_rows(int n, const std::function<void(int, int)>& f) {
  int nth = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
  if (n < 20000) nth = 1;
  if (nth == 1) {
    f(0, n);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() { f((int)((int64_t)n * t / nth), (int)((int64_t)n * (t + 1) / nth)); });
  for (auto& x : th) x.join();
}

// Transpose of a matrix whose rows [rs[b], rs[b+1]) only reference columns [cs[b], cs[b+1]) (the transfer operators
// are block diagonal by subdomain): the blocks are transposed independently on host threads.
static HostCsr transpose_blocks(const HostCsr& a, int ncols, const std::vector<int>& rs, const std::vector<int>& cs) {
  HostCsr t;
  t.n = ncols;
  t.rowptr.assign(ncols + 1, 0);
  for (int c : a.col) t.rowptr[c + 1]++;
  for (int i = 0; i < ncols; ++i) t.rowptr[i + 1] += t.rowptr[i];
  t.col.resize(a.col.size());
  t.val.resize(a.val.size());
  const int nb = (int)rs.size() - 1;
  auto one = [&](int b) {
    std::vector<int> fill(t.rowptr.begin() + cs[b], t.rowptr.begin() + cs[b + 1]);
    for (int i = rs[b]; i < rs[b + 1]; ++i)
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        const int p = fill[a.col[k] - cs[b]]++;
        t.col[p] = i;
        t.val[p] = a.val[k];
      }
  };
  std::vector<std::thread> th;
  for (int b = 0; b < nb; ++b) th.emplace_back(one, b);
  for (auto& x : th) x.join();
  return t;
}

static HostCsr transpose(const HostCsr& a, int ncols) {
  HostCsr t;
  t.n = ncols;
  t.rowptr.assign(ncols + 1, 0);
  for (int c : a.col) t.rowptr[c + 1]++;
  for (int i = 0; i < ncols; ++i) t.rowptr[i + 1] += t.rowptr[i];
  t.col.resize(a.col.size());
  t.val.resize(a.val.size());
  std::vector<int> fill(t.rowptr.begin(), t.rowptr.end() - 1);
  for (int i = 0; i < a.n; ++i)
    for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
      const int p = fill[a.col[k]]++;
      t.col[p] = i;
      t.val[p] = a.val[k];
    }
  return t;
}

// C = A * B, two passes (symbolic row counts, then numeric fill into preallocated arrays) with one
// marker array per thread; rows are split into contiguous ranges over host threads; columns unsorted.
static HostCsr spgemm(const HostCsr& a, const HostCsr& b, int ncols_b) {
  const int n = a.n;
  HostCsr c;
  c.n = n;
  c.rowptr.assign(n + 1, 0);
  int nth = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
  if (a.col.size() < 200000) nth = 1;   // by work, not by rows: a restriction operator has few, long rows
  nth = std::max(1, std::min(nth, n));
  // contiguous row ranges of (about) equal nnz
  std::vector<int> cut(nth + 1, n);
  cut[0] = 0;
  for (int t = 1; t < nth; ++t) {
    const int64_t target = (int64_t)a.rowptr[n] * t / nth;
    cut[t] = (int)(std::lower_bound(a.rowptr.begin(), a.rowptr.begin() + n + 1, (int)target) - a.rowptr.begin());
    cut[t] = std::min(n, std::max(cut[t], cut[t - 1]));
  }
  auto range = [&](int t, int& r0, int& r1) {
    r0 = cut[t];
    r1 = cut[t + 1];
  };
  auto run = [&](const std::function<void(int)>& f) {
    if (nth == 1) { f(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t) th.emplace_back(f, t);
    for (auto& x : th) x.join();
  };
  run([&](int t) {  // symbolic
    int r0, r1;
    range(t, r0, r1);
    std::vector<int> marker(ncols_b, -1);
    for (int i = r0; i < r1; ++i) {
      int cnt = 0;
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        const int j = a.col[k];
        for (int l = b.rowptr[j]; l < b.rowptr[j + 1]; ++l)
          if (marker[b.col[l]] != i) { marker[b.col[l]] = i; ++cnt; }
      }
      c.rowptr[i + 1] = cnt;
    }
  });
  for (int i = 0; i < n; ++i) c.rowptr[i + 1] += c.rowptr[i];
  c.col.resize(c.rowptr[n]);
  c.val.resize(c.rowptr[n]);
  run([&](int t) {  // numeric
    int r0, r1;
    range(t, r0, r1);
    std::vector<int> marker(ncols_b, -1);
    for (int i = r0; i < r1; ++i) {
      const int base = c.rowptr[i];
      int cnt = 0;
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        const int j = a.col[k];
        const double av = a.val[k];
        for (int l = b.rowptr[j]; l < b.rowptr[j + 1]; ++l) {
          const int cc = b.col[l];
          if (marker[cc] < base) {
            marker[cc] = base + cnt;
            c.col[base + cnt] = cc;
            c.val[base + cnt] = av * b.val[l];
            ++cnt;
          } else {
            c.val[marker[cc]] += av * b.val[l];
          }
        }
      }
    }
  });
  return c;
}

// Vanek-style aggregation inside one diagonal block [r0, r1): returns #aggregates, agg[i] local ids.
// theta > 0: only STRONG connections |a_ij| >= theta sqrt(a_ii a_jj) tie nodes together (Vanek, Mandel, Brezina 1996).
//
// Round 3: the greedy seeding pass (phase 1) is inherently serial, everything around it is not.  The strong neighbour
// lists are gathered first, in row ranges on `nthreads` host threads (one pass over the block's entries: the diagonal
// and the strength test read 12 bytes per entry, the serial pass then walks 4 bytes per STRONG entry -- a coarse Galerkin
// operator of 31 entries per row keeps about a quarter of them), and the leftovers' choice of an aggregate (phase 2, which
// reads the aggregates of phase 1 and writes only its own row) runs in row ranges as well.  Same predicate, same visiting
// order, same tie-breaking: the aggregates are the ones the serial loops produced (tests: iteration counts unchanged).
// One 869 k-row block of 26.5 M entries (first coarse level of one rank of 368^3): 0.07 -> 0.02 s.
static int aggregate_block(const HostCsr& a, int r0, int r1, std::vector<int>& agg, double theta = 0.0, int nthreads = 1) {
  int na = 0;
  const int nb = r1 - r0;
  if (nb <= 0) return 0;
  {   // test switches: GENEO_AGG_THREADS forces the thread count, GENEO_AGG_MIN_NNZ the size below which one thread does it all
    const char* ft = getenv("GENEO_AGG_THREADS");
    const char* fm = getenv("GENEO_AGG_MIN_NNZ");
    if (ft) nthreads = atoi(ft);
    if ((int64_t)a.rowptr[r1] - a.rowptr[r0] < (fm ? atoll(fm) : 4000000LL)) nthreads = 1;
    nthreads = std::max(1, std::min(nthreads, ft ? nb : nb / 4096 + 1));
  }
  auto ranges = [&](const std::function<void(int, int, int)>& f) {       // f(thread, row begin, row end)
    if (nthreads == 1) { f(0, r0, r1); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t)
      th.emplace_back([&, t]() { f(t, r0 + (int)((int64_t)nb * t / nthreads), r0 + (int)((int64_t)nb * (t + 1) / nthreads)); });
    f(0, r0, r0 + (int)((int64_t)nb / nthreads));
    for (auto& x : th) x.join();
  };
  std::vector<double> dg;
  if (theta > 0.0) {
    dg.assign(nb, 0.0);
    ranges([&](int, int b, int e) {
      for (int i = b; i < e; ++i) {
        agg[i] = -1;
        for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k)
          if (a.col[k] == i) dg[i - r0] = std::fabs(a.val[k]);
      }
    });
  } else {
    ranges([&](int, int b, int e) { for (int i = b; i < e; ++i) agg[i] = -1; });
  }
  const double t2 = theta * theta;
  auto strong = [&](int i, int k) {
    const double v = a.val[k];
    if (v == 0.0) return false;
    if (theta <= 0.0) return true;
    const int j = a.col[k];
    if (j < r0 || j >= r1) return false;
    return v * v >= t2 * dg[i - r0] * dg[j - r0];
  };
  // strong neighbour lists (diagonal included when it passes the test, as the loops below always treated it), per range
  // (scnt, join: every entry that is read has been written by the passes below -- no serial fill of 26 MB arrays, which
  // together with the serial tail loops were 0.05 of the 0.08 s a 6.5 M-row block took, 0.047 s of it on the critical path
  // of one rank of 368^3: the seeding pass itself is 0.015 s)
  // theta = 0 (mesh matrices, LOBPCG's hierarchy): every non-zero entry is strong -- the seeding pass reads the matrix
  // itself, and only the rows of nodes that are still free (no list to build: one pass over 45 M entries less)
  // (the serial pass is the critical resource: it should not read the values -- they only matter when the block stores
  // explicit zeros, which a parallel scan rules out first; with zeros stored the lists are built as for theta > 0)
  bool direct = (theta <= 0.0);
  if (direct) {
    std::vector<char> zero_in(nthreads, 0);
    ranges([&](int t, int b, int e) {
      const double* v = a.val.data();
      char z = 0;
      for (int64_t k = a.rowptr[b]; k < a.rowptr[e]; ++k) z |= (v[k] == 0.0);
      zero_in[t] = z;
    });
    for (int t = 0; t < nthreads; ++t)
      if (zero_in[t]) direct = false;
  }
  std::vector<std::vector<int>> scol(nthreads);
  std::unique_ptr<int[]> scnt_own(new int[direct ? 1 : nb]);
  int* scnt = scnt_own.get();
  if (!direct) ranges([&](int t, int b, int e) {
    std::vector<int>& out = scol[t];
    out.reserve((size_t)(a.rowptr[e] - a.rowptr[b]) / (theta > 0.0 ? 3 : 1) + 16);
    for (int i = b; i < e; ++i) {
      int c = 0;
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k)
        if (strong(i, k)) { out.push_back(a.col[k]); ++c; }
      scnt[i - r0] = c;
    }
  });
  const auto t_ph1 = std::chrono::high_resolution_clock::now();
  if (direct) {                           // phase 1 on the matrix rows: same visiting order, same predicate
    for (int i = r0; i < r1; ++i) {
      if (agg[i] >= 0) continue;
      const int ka = a.rowptr[i], kb = a.rowptr[i + 1];
      bool ok = true;
      for (int k = ka; k < kb; ++k) {
        const int j = a.col[k];
        if (j != i && agg[j] >= 0) { ok = false; break; }
      }
      if (ok) {
        for (int k = ka; k < kb; ++k) agg[a.col[k]] = na;
        agg[i] = na++;
      }
    }
  } else
  for (int t = 0; t < nthreads; ++t) {   // phase 1: a free node whose whole (strong) neighbourhood is free seeds an aggregate
    const int b = (t == 0) ? r0 : r0 + (int)((int64_t)nb * t / nthreads);
    const int e = (nthreads == 1) ? r1 : r0 + (int)((int64_t)nb * (t + 1) / nthreads);
    const int* sp = scol[t].data();
    for (int i = b; i < e; ++i) {
      const int c = scnt[i - r0];
      if (agg[i] < 0) {
        bool ok = true;
        for (int k = 0; k < c; ++k) {
          const int j = sp[k];
          if (j != i && agg[j] >= 0) { ok = false; break; }
        }
        if (ok) {
          for (int k = 0; k < c; ++k) agg[sp[k]] = na;
          agg[i] = na++;
        }
      }
      sp += c;
    }
  }
  if (getenv("GENEO_DEBUG") && nb > 500000)
    fprintf(stderr, "[amg/host] aggregation of a %d-row block on %d thread(s): serial seeding pass %.3f s\n", nb, nthreads,
            std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t_ph1).count());
  std::unique_ptr<int[]> join_own(new int[nb]);
  int* join = join_own.get();
  std::vector<int64_t> left(nthreads, 0);       // nodes without an aggregate after phase 2, per range
  ranges([&](int t, int b, int e) {      // phase 2: leftovers join the most strongly connected aggregate (of phase 1)
    for (int i = b; i < e; ++i) {
      join[i - r0] = -1;
      if (agg[i] >= 0) continue;
      double best = 0.0;
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        const int j = a.col[k];
        if (j != i && agg[j] >= 0 && std::fabs(a.val[k]) > best) {
          best = std::fabs(a.val[k]);
          join[i - r0] = agg[j];
        }
      }
    }
  });
  // (phase 2 reads the aggregates of phase 1 only: the assignments wait until every range is through)
  ranges([&](int t, int b, int e) {
    int64_t l = 0;
    for (int i = b; i < e; ++i)
      if (agg[i] < 0) {
        if (join[i - r0] >= 0) agg[i] = join[i - r0];
        else ++l;
      }
    left[t] = l;
  });
  int64_t nleft = 0;
  for (int t = 0; t < nthreads; ++t) nleft += left[t];
  if (nleft)
    for (int i = r0; i < r1; ++i)  // phase 3: isolated nodes, numbered in row order
      if (agg[i] < 0) agg[i] = na++;
  return na;
}
// host threads one block's aggregation may use when `nblocks` of them run side by side
static int aggregate_threads(int nblocks) {
  const int hw = std::max(1, (int)std::thread::hardware_concurrency());
  return std::max(1, std::min(16, hw / std::max(1, nblocks)));
}

// Strength threshold of level l.  Level 0 keeps every entry (the operators handed in are mesh matrices: an M-matrix
// stencil has no weak entries worth dropping, and a high-contrast one is the eigensolver's business); from level 1 on
// the Galerkin operators of smoothed aggregation carry many weak far couplings (27-point-like rows of 31 entries on a
// 7-point problem) and distance-1 aggregates over ALL of them coarsen 44-fold -- an interpolation the next level
// cannot support.  theta_l = theta 0.5^l (Vanek, Mandel, Brezina).
static double amg_strength(const AmgParams& prm, int level) {
  if (level < 1 || prm.strength <= 0.0) return 0.0;
  return prm.strength * std::pow(0.5, level);
}

// Jacobi scaling and Gershgorin bound of D^-1 A, PER SUBDOMAIN BLOCK.  Every use of dinv in the hierarchy is a smoothing
// step w D^-1 with a weight w proportional to 1 / rho (prolongator smoothing 4 / (3 rho), damped Jacobi, Chebyshev), and the
// kernels take ONE scalar w per level: the level keeps rho = the largest block's bound and a block with a smaller bound
// rho_s gets its rows of dinv scaled by rho / rho_s, so that w dinv is that block's own (c / rho_s) D^-1.  A subdomain's
// hierarchy then does not depend on which other subdomains share the batch: the same V-cycle whether the rank holds
// eight subdomains or one (bench.py --scaling strong: N = 1, 2, 4, 8), or the eigensolves run group by group
// (PC::eigen_grouped).  Blocks that attain the maximum (every block of the fine-level Laplacian) are scaled by exactly 1.
static double gershgorin_rho(const HostCsr& a, std::vector<double>& dinv, const std::vector<int>& suboff) {
  dinv.assign(a.n, 1.0);
  const int nsub = std::max(1, (int)suboff.size() - 1);
  std::vector<double> rho_s(nsub, 0.0);
  bool bad = false;
  std::mutex mu;
  parallel_rows(a.n, [&](int r0, int r1) {
    bool neg = false;
    int sd = suboff.size() > 1 ? (int)(std::upper_bound(suboff.begin(), suboff.end(), r0) - suboff.begin()) - 1 : 0;
    std::vector<std::pair<int, double>> mine;      // (subdomain, bound over this range's rows of it)
    double rho = 0.0;
    for (int i = r0; i < r1; ++i) {
      if (suboff.size() > 1 && i >= suboff[sd + 1]) {
        mine.emplace_back(sd, rho);
        rho = 0.0;
        while (i >= suboff[sd + 1]) ++sd;
      }
      double d = 0.0, row = 0.0;
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        row += std::fabs(a.val[k]);
        if (a.col[k] == i) d += a.val[k];
      }
      if (!(d > 0.0)) { neg = true; continue; }
      dinv[i] = 1.0 / d;
      rho = std::max(rho, row / d);
    }
    if (r1 > r0) mine.emplace_back(sd, rho);
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : mine) rho_s[std::min(kv.first, nsub - 1)] = std::max(rho_s[std::min(kv.first, nsub - 1)], kv.second);
    bad = bad || neg;
  });
  if (bad) throw std::runtime_error("AMG: non-positive diagonal");
  double rho = 0.0;
  for (double v : rho_s) rho = std::max(rho, v);
  if (!(rho > 0)) return 2.0;
  static const bool shared = getenv("GENEO_AMG_SHARED_RHO") != nullptr;   // A/B: the rounds-1..3 rule (one bound per level)
  if (suboff.size() > 2 && !shared) {
    for (int s = 0; s < nsub; ++s) {
      if (!(rho_s[s] > 0.0) || rho_s[s] == rho) continue;
      const double f = rho / rho_s[s];
      parallel_rows(suboff[s + 1] - suboff[s], [&](int r0, int r1) {
        for (int i = suboff[s] + r0; i < suboff[s] + r1; ++i) dinv[i] *= f;
      });
    }
  }
  return rho;
}

// dense inverse of every coarsest block (SPD: Cholesky; tiny pivots regularised)
static void coarse_inverse(const HostCsr& CA, const std::vector<int>& csuboff, std::vector<double>& coarse_inv,
                           std::vector<int64_t>& coarse_base) {
  const int nsub = (int)csuboff.size() - 1;
  struct { const std::vector<int>& suboff; } C{csuboff};
  coarse_base.assign(nsub + 1, 0);
  for (int s = 0; s < nsub; ++s) {
    const int64_t m = C.suboff[s + 1] - C.suboff[s];
    coarse_base[s + 1] = coarse_base[s] + m * m;
  }
  coarse_inv.assign((size_t)std::max<int64_t>(1, coarse_base[nsub]), 0.0);
  auto invert = [&](int s) {
    const int r0 = C.suboff[s], m = C.suboff[s + 1] - r0;
    if (m == 0) return;
    std::vector<double> a((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i)
      for (int k = CA.rowptr[r0 + i]; k < CA.rowptr[r0 + i + 1]; ++k) a[(size_t)i * m + (CA.col[k] - r0)] += CA.val[k];
    for (int i = 0; i < m; ++i)
      for (int j = i + 1; j < m; ++j) a[(size_t)i * m + j] = a[(size_t)j * m + i] = 0.5 * (a[(size_t)i * m + j] + a[(size_t)j * m + i]);
    // Singular blocks (a floating subdomain's Neumann matrix when the element matrices carry no mass term, --inpEps 0)
    // get the reference's treatment of its MUMPS factors (tuneSolver, geneo.cpp:76-92): null pivots are detected and
    // pinned, the V-cycle then applies a bounded generalised inverse instead of amplifying the kernel component of
    // every residual by 1 / rounding.
    std::vector<double> l = a;
    int fixed = dense::cholesky_fix_null_pivots(l, m);
    if (fixed < 0) {
      // A Galerkin block that is positive semi-definite by construction may come out indefinite by a rounding error
      // larger than the null-pivot window: one retry with a trace shift of 1e-10 (a preconditioner's coarsest solve:
      // the outer PCG absorbs it) before the block is declared indefinite.
      double tr = 0.0;
      for (int i = 0; i < m; ++i) tr += std::fabs(a[(size_t)i * m + i]);
      l = a;
      for (int i = 0; i < m; ++i) l[(size_t)i * m + i] += 1e-10 * tr / m;
      fixed = dense::cholesky_fix_null_pivots(l, m);
    }
    if (fixed < 0) throw std::runtime_error("AMG: coarsest block is not positive semi-definite");
    if (fixed > 0) g_null_pivots.fetch_add(fixed);
    std::vector<double> e(m);
    double* inv = coarse_inv.data() + coarse_base[s];
    for (int j = 0; j < m; ++j) {
      std::fill(e.begin(), e.end(), 0.0);
      e[j] = 1.0;
      dense::cholesky_solve(l, m, e.data());
      for (int i = 0; i < m; ++i) inv[(size_t)i * m + j] = e[i];
    }
  };
  {
    const int nth = std::max(1, std::min(nsub, std::min(16, (int)std::thread::hardware_concurrency())));
    std::vector<std::thread> th;
    std::exception_ptr err;      // an indefinite block must come back as an error of the set-up, not std::terminate
    std::mutex emu;
    for (int t = 0; t < nth; ++t)
      th.emplace_back([&, t]() {
        try {
          for (int s = t; s < nsub; s += nth) invert(s);
        } catch (...) {
          std::lock_guard<std::mutex> lk(emu);
          if (!err) err = std::current_exception();
        }
      });
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
  }
}

void amg_setup_host(const HostCsr& A, const std::vector<int>& suboff, const AmgParams& prm,
                    std::vector<AmgLevelHost>& levels, std::vector<double>& coarse_inv,
                    std::vector<int64_t>& coarse_base) {
  levels.clear();
  const int nsub = (int)suboff.size() - 1;
  levels.emplace_back();
  levels.back().n = A.n;
  levels.back().nnz = A.val.size();
  levels.back().suboff = suboff;
  while (true) {
    AmgLevelHost& L = levels.back();
    const HostCsr& LA = (levels.size() == 1) ? A : L.A;   // level 0: the caller's matrix, not a copy
    L.rho = gershgorin_rho(LA, L.dinv, L.suboff);
    int maxblk = 0;
    for (int s = 0; s < nsub; ++s) maxblk = std::max(maxblk, L.suboff[s + 1] - L.suboff[s]);
    if (maxblk <= prm.coarse_size || (int)levels.size() >= prm.max_levels) break;
    // aggregation per subdomain block
    const bool dbg = getenv("GENEO_DEBUG") != nullptr;
    auto tnow = []() { return std::chrono::high_resolution_clock::now(); };
    auto tsec = [](std::chrono::high_resolution_clock::time_point a, std::chrono::high_resolution_clock::time_point b) {
      return std::chrono::duration<double>(b - a).count();
    };
    auto t_0 = tnow();
    const int n = LA.n;
    std::vector<int> agg(n, -1), csub(nsub + 1, 0);
    {  // the diagonal blocks are independent: one host thread per subdomain
      std::vector<int> nagg(nsub, 0);
      std::vector<std::thread> th;
      for (int s = 0; s < nsub; ++s)
        th.emplace_back([&, s]() { nagg[s] = aggregate_block(LA, L.suboff[s], L.suboff[s + 1], agg, amg_strength(prm, (int)levels.size() - 1), aggregate_threads(nsub)); });
      for (auto& x : th) x.join();
      for (int s = 0; s < nsub; ++s) csub[s + 1] = csub[s] + nagg[s];
      for (int s = 0; s < nsub; ++s)
        for (int i = L.suboff[s]; i < L.suboff[s + 1]; ++i) agg[i] += csub[s];
    }
    const int nc = csub[nsub];
    if (nc >= n) break;  // no coarsening possible
    // tentative prolongator (piecewise constant) and its Jacobi smoothing  P = (I - w D^-1 A) P0
    HostCsr P0;
    P0.n = n;
    P0.rowptr.resize(n + 1);
    P0.col.resize(n);
    P0.val.assign(n, 1.0);
    for (int i = 0; i <= n; ++i) P0.rowptr[i] = i;
    for (int i = 0; i < n; ++i) P0.col[i] = agg[i];
    const double omega = 4.0 / (3.0 * L.rho);
    auto t_1 = tnow();
    HostCsr AP0 = spgemm(LA, P0, nc);
    auto t_2 = tnow();
    HostCsr P;
    P.n = n;
    P.rowptr.assign(n + 1, 0);
    parallel_rows(n, [&](int r0, int r1) {  // row sizes: the entries of A*P0 plus the tentative one if absent
      for (int i = r0; i < r1; ++i) {
        bool has = false;
        for (int k = AP0.rowptr[i]; k < AP0.rowptr[i + 1]; ++k)
          if (AP0.col[k] == agg[i]) has = true;
        P.rowptr[i + 1] = (AP0.rowptr[i + 1] - AP0.rowptr[i]) + (has ? 0 : 1);
      }
    });
    for (int i = 0; i < n; ++i) P.rowptr[i + 1] += P.rowptr[i];
    P.col.resize(P.rowptr[n]);
    P.val.resize(P.rowptr[n]);
    parallel_rows(n, [&](int r0, int r1) {
      for (int i = r0; i < r1; ++i) {
        int q = P.rowptr[i];
        bool has = false;
        for (int k = AP0.rowptr[i]; k < AP0.rowptr[i + 1]; ++k, ++q) {
          double v = -omega * L.dinv[i] * AP0.val[k];
          if (AP0.col[k] == agg[i]) { v += 1.0; has = true; }
          P.col[q] = AP0.col[k];
          P.val[q] = v;
        }
        if (!has) {
          P.col[q] = agg[i];
          P.val[q] = 1.0;
        }
      }
    });
    auto t_3 = tnow();
    HostCsr R = (nsub > 1 && n >= 20000) ? transpose_blocks(P, nc, L.suboff, csub) : transpose(P, nc);
    auto t_4 = tnow();
    HostCsr AP = spgemm(LA, P, nc);
    auto t_5 = tnow();
    HostCsr Ac = spgemm(R, AP, nc);
    auto t_6 = tnow();
    if (dbg)
      fprintf(stderr, "[amg] level n %d -> %d nnz %zu -> %zu | agg %.3f AP0 %.3f P %.3f Rt %.3f AP %.3f RAP %.3f s\n", n, nc,
              LA.val.size(), Ac.val.size(), tsec(t_0, t_1), tsec(t_1, t_2), tsec(t_2, t_3), tsec(t_3, t_4),
              tsec(t_4, t_5), tsec(t_5, t_6));
    L.P = std::move(P);
    L.R = std::move(R);
    levels.emplace_back();
    levels.back().n = Ac.n;
    levels.back().nnz = Ac.val.size();
    levels.back().A = std::move(Ac);
    levels.back().suboff = csub;
  }
  coarse_inverse((levels.size() == 1) ? A : levels.back().A, levels.back().suboff, coarse_inv, coarse_base);
}

// ------------------------------------------------------------------------------ device V-cycle
AmgDevice::~AmgDevice() { free_all(); }

void AmgDevice::free_all() {
  for (auto& L : lv) {
    bk::csr_free(L.Acs);
    bk::csr_free(L.M);
    if (L.own_A) bk::csr_free(L.A);
    else if (L.own_lp_A) bk::csr_free_lp(L.A);    // the companion of a borrowed matrix, when it was made here
    bk::csr_free(L.P);
    bk::csr_free(L.R);
    bk::dfree(L.dinv); bk::dfree(L.b); bk::dfree(L.x); bk::dfree(L.r); bk::dfree(L.d); bk::dfree(L.ad);
  }
  lv.clear();
  if (cch.start) bk::chunks_free(cch);
  bk::dfree(d_inv);
  bk::dfree(d_invbase);
  d_inv = nullptr;
  d_invbase = nullptr;
}

void AmgDevice::upload(const std::vector<AmgLevelHost>& levels, const std::vector<double>& coarse_inv,
                       const std::vector<int64_t>& coarse_base, const AmgParams& p, int max_m, const bk::Csr* fine_dev) {
  free_all();
  prm = p;
  maxm = std::max(1, max_m);
  double nnz0 = 0.0, nnzt = 0.0;
  for (size_t l = 0; l < levels.size(); ++l) {
    const AmgLevelHost& H = levels[l];
    Lvl L;
    L.n = H.n;
    L.rho = H.rho;
    if (l == 0 && fine_dev) {
      L.A = *fine_dev;
      L.own_A = false;
    } else {
      if (l == 0) throw std::runtime_error("AMG: the level-0 matrix must already be resident (fine_dev)");
      L.A = bk::csr_upload(H.A.n, H.A.rowptr.data(), H.A.col.data(), H.A.val.data());
    }
    if (l + 1 < levels.size()) {
      L.P = bk::csr_upload(H.P.n, H.P.rowptr.data(), H.P.col.data(), H.P.val.data());
      // R = P^T is transposed on the device (a third of this level's PCIe traffic); the host's copy is the fallback
      // when a row of P^T exceeds the kernel's capacity
      bool ok = !getenv("GENEO_AMG_UPLOAD_R");
      if (ok) {
        L.R = bk::transpose(L.P, H.R.n, &ok);
        if (ok) bk::csr_finish(L.R);
      }
      if (!ok) L.R = bk::csr_upload(H.R.n, H.R.rowptr.data(), H.R.col.data(), H.R.val.data());
    }
    L.dinv = (double*)bk::alloc(sizeof(double) * std::max(1, L.n));
    bk::h2d(L.dinv, H.dinv.data(), sizeof(double) * L.n);
    alloc_level_buffers(L, l > 0);
    L.fused = bk::csr_fusable(L.A) && (l + 1 == (int)levels.size() || bk::csr_fusable(L.P)) && !getenv("GENEO_AMG_UNFUSED");
    if (l + 1 < levels.size()) {
      make_column_scaled(L);
      make_post_matrix(L, nullptr, levels[l + 1].n);
    }
    if (l == 0) nnz0 = (double)H.nnz;
    nnzt += (double)H.nnz;
    L.suboff = H.suboff;
    lv.push_back(L);
  }
  opc = nnz0 > 0 ? nnzt / nnz0 : 1.0;
  const AmgLevelHost& C = levels.back();
  cch = bk::chunks_upload((int)C.suboff.size() - 1, C.suboff.data());
  d_inv = (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, coarse_inv.size()));
  bk::h2d(d_inv, coarse_inv.data(), sizeof(double) * coarse_inv.size());
  d_invbase = (int64_t*)bk::alloc(sizeof(int64_t) * coarse_base.size());
  bk::h2d(d_invbase, coarse_base.data(), sizeof(int64_t) * coarse_base.size());
  if (prm.single) make_single();
}

// Single-precision companions (float values, 16-bit column offsets per slice: 6 bytes per entry instead of 12) of every
// level matrix on the sliced path.  Only the single-vector V-cycle reads them -- the preconditioner of the FP64 PCG of
// the local solves, whose passes over the fine and first coarse matrices are HBM streams; its arithmetic and its
// vectors stay FP64, and so do the block cycles of LOBPCG (their traffic is the vector blocks, not the matrix).
void AmgDevice::make_single() {
  nlp = 0;
  for (Lvl& L : lv) {
    if (!L.fused) continue;
    // the level-0 matrix may be borrowed: the companion then hangs off OUR copy of the descriptor
    const bool had = bk::csr_has_lp(L.A);
    if (bk::csr_make_lp(L.A)) ++nlp;
    L.own_lp_A = !L.own_A && !had && bk::csr_has_lp(L.A);
    if (L.Acs.n && bk::csr_make_lp(L.Acs, &L.A)) ++nlp;
    if (L.P.n && bk::csr_make_lp(L.P)) ++nlp;
    if (L.R.n && bk::csr_make_lp(L.R)) ++nlp;
    if (L.M.n && bk::csr_make_lp(L.M)) ++nlp;
  }
}

double AmgDevice::jacobi_weight(const Lvl& L) const {
  const double lmax = 1.1 * L.rho, lmin = lmax / std::max(1.5, prm.smooth_ratio);
  return 1.0 / (0.5 * (lmax + lmin));
}

// The post-smoothing half of the damped-Jacobi V-cycle in one product.  With x1 = w D^-1 b and r1 = b - A x1 from the
// zero-guess sweep (EPI_PRE) and e the coarse correction:
//     t = x1 + P e ,  x = t + w D^-1 (b - A t)   =   x1 + w D^-1 r1 + (P - w D^-1 A P) e   =   w D^-1 (b + r1) + M e
// so the prolongation, the correction and the sweep cost ONE pass over M = P - w D^-1 A P (about the entries of A, and
// it gathers the small coarse vector, not a fine one) instead of a pass over P and a pass over A with a fine gather:
// three fine-vector passes (b, r1 in, x out) instead of five, and the zero-guess sweep stores r1 only.  A P is a by-product of the Galerkin product (device
// set-up) or one more device sparse product (host set-up).  `ap` (consumed) may be null.
void AmgDevice::make_post_matrix(Lvl& L, bk::Csr* ap, int nc) {
  const bool want = prm.smooth_degree <= 1 && L.fused && !getenv("GENEO_AMG_NO_POST_MATRIX");
  bk::Csr AP;
  bool ok = true;
  if (ap) AP = *ap;
  else if (want) AP = bk::spgemm(L.A, L.P, nc, &ok);
  if (!want || !ok || AP.n == 0) {
    if (AP.n) bk::csr_free(AP);
    return;
  }
  if (!bk::post_matrix(AP, L.P, L.dinv, jacobi_weight(L))) {
    bk::csr_free(AP);
    return;
  }
  bk::csr_finish(AP);
  if (!bk::csr_fusable(AP)) {     // long rows: stay with the two-launch form
    bk::csr_free(AP);
    return;
  }
  L.M = AP;
}

void AmgDevice::make_column_scaled(Lvl& L) {
  // only the sliced kernels know about pre-scaled values; ragged levels keep the explicit scaling
  if (L.fused && L.A.vec_lpr == 0 && L.A.nlong == 0 && L.A.n > 0 && !getenv("GENEO_AMG_NO_PRESCALE"))
    L.Acs = bk::csr_scaled_alias(L.A, nullptr, L.dinv, true);
}

void AmgDevice::alloc_level_buffers(Lvl& L, bool coarse) {
  const size_t blk = sizeof(double) * std::max<size_t>(1, (size_t)L.n * maxm);
  L.r = (double*)bk::alloc(blk);
  L.d = (double*)bk::alloc(blk);
  if (prm.smooth_degree > 1) L.ad = (double*)bk::alloc(blk);
  if (coarse) {
    L.b = (double*)bk::alloc(blk);
    L.x = (double*)bk::alloc(blk);
  }
}

AmgLevelHostPart amg_level_host_part(const HostCsr& Ah, const std::vector<int>& so, const AmgParams& prm, int l) {
  AmgLevelHostPart h;
  const int nsub = (int)so.size() - 1;
  const auto t_hp0 = std::chrono::high_resolution_clock::now();
  h.rho = gershgorin_rho(Ah, h.dinv, so);
  const auto t_hp1 = std::chrono::high_resolution_clock::now();
  int maxblk = 0;
  for (int s = 0; s < nsub; ++s) maxblk = std::max(maxblk, so[s + 1] - so[s]);
  h.last = (maxblk <= prm.coarse_size || l + 1 >= prm.max_levels);
  h.csub.assign(nsub + 1, 0);
  if (!h.last) {  // aggregation per subdomain block, on the host
    const int n = Ah.n;
    h.agg.assign(n, -1);
    std::vector<int> nagg(nsub, 0);
    std::vector<std::thread> th;
    for (int s = 0; s < nsub; ++s)
      th.emplace_back([&, s]() { nagg[s] = aggregate_block(Ah, so[s], so[s + 1], h.agg, amg_strength(prm, l), aggregate_threads(nsub)); });
    for (auto& x : th) x.join();
    for (int s = 0; s < nsub; ++s) h.csub[s + 1] = h.csub[s] + nagg[s];
    for (int s = 1; s < nsub; ++s) {       // (block 0 keeps its numbers)
      const int off = h.csub[s];
      parallel_rows(so[s + 1] - so[s], [&](int a0, int a1) {
        for (int i = so[s] + a0; i < so[s] + a1; ++i) h.agg[i] += off;
      });
    }
    h.nc = h.csub[nsub];
    if (h.nc >= n) h.last = true;
  }
  if (getenv("GENEO_DEBUG") && Ah.n > 500000)
    fprintf(stderr, "[amg/host] level %d, %d rows: diagonal + bound %.3f s, aggregation %.3f s\n", l, Ah.n,
            std::chrono::duration<double>(t_hp1 - t_hp0).count(),
            std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t_hp1).count());
  return h;
}

// Host arrays of the coarse matrices that build_on_device downloads, kept from one hierarchy to the next: a fresh
// std::vector of the first coarse matrix (126^3: 10 M entries, 123 MB) costs 20 ms of page faults and zero fill on the
// thread that sizes it -- the device sat idle for exactly that long in every set-up (kernel timeline, round 3).  Arrays
// returned to the pool keep their size; taking one of the same size (the next set-up of the same problem) touches nothing.
namespace {
struct HostCsrPool {
  std::mutex mu;
  std::vector<HostCsr> parked;
  static constexpr size_t MAX_PARKED = 12;
  HostCsr take(int n, size_t nnz) {
    HostCsr out;
    {
      std::lock_guard<std::mutex> lk(mu);
      int best = -1;
      for (size_t i = 0; i < parked.size(); ++i) {
        const HostCsr& h = parked[i];
        if (h.col.size() == nnz && h.rowptr.size() == (size_t)n + 1) { best = (int)i; break; }     // same problem again
        if (h.col.capacity() >= nnz && h.rowptr.capacity() >= (size_t)n + 1 &&
            (best < 0 || h.col.capacity() < parked[best].col.capacity()))
          best = (int)i;
      }
      if (best >= 0) {
        out = std::move(parked[best]);
        parked.erase(parked.begin() + best);
      }
    }
    out.n = n;
    out.rowptr.resize((size_t)n + 1);
    out.col.resize(nnz);
    out.val.resize(nnz);
    return out;
  }
  void give(HostCsr&& h) {
    if (h.col.capacity() < (1u << 16)) return;       // small ones are not worth keeping
    std::lock_guard<std::mutex> lk(mu);
    if (parked.size() >= MAX_PARKED) {               // drop the smallest
      size_t k = 0;
      for (size_t i = 1; i < parked.size(); ++i)
        if (parked[i].col.capacity() < parked[k].col.capacity()) k = i;
      if (parked[k].col.capacity() >= h.col.capacity()) return;
      parked.erase(parked.begin() + k);
    }
    parked.push_back(std::move(h));
  }
};
HostCsrPool g_host_csr_pool;
}  // namespace

bool AmgDevice::build_on_device(const HostCsr& A, const std::vector<int>& suboff, const AmgParams& p, int max_m,
                                const bk::Csr* fine_dev, const AmgLevelHostPart* level0) {
  free_all();
  prm = p;
  maxm = std::max(1, max_m);
  const int nsub = (int)suboff.size() - 1;
  const bool dbg = getenv("GENEO_DEBUG") != nullptr;
  auto tnow = []() { return std::chrono::high_resolution_clock::now(); };
  auto tsec = [](std::chrono::high_resolution_clock::time_point a, std::chrono::high_resolution_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  HostCsr Acur;                       // host copy of the current level's matrix (levels >= 1)
  AmgLevelHostPart hp_next;           // host part of the level to come, when the helper thread of the last one made it
  bool have_next = false;
  std::vector<int> so = suboff;
  double nnz0 = 0.0, nnzt = 0.0;
  bk::Csr Adev = *fine_dev;
  bool own = false;
  auto abandon = [&]() {
    if (own) bk::csr_free(Adev);
    free_all();
    return false;
  };
  for (int l = 0;; ++l) {
    const HostCsr& Ah = (l == 0) ? A : Acur;
    auto t_0 = tnow();
    Lvl L;
    L.n = Ah.n;
    L.A = Adev;
    L.own_A = own;
    L.suboff = so;
    // host part of the level (diagonal, Gershgorin bound, aggregates): level 0 may have been computed early by the caller
    AmgLevelHostPart hp_own;
    if (l == 0 && level0) {
      // computed early by the caller
    } else if (have_next) {
      hp_own = std::move(hp_next);        // computed by the previous level's helper thread, behind its device work
      have_next = false;
    } else {
      hp_own = amg_level_host_part(Ah, so, prm, l);
    }
    const AmgLevelHostPart& hp = (l == 0 && level0) ? *level0 : hp_own;
    const std::vector<double>& dinv = hp.dinv;
    L.rho = hp.rho;
    L.dinv = (double*)bk::alloc(sizeof(double) * std::max(1, L.n));
    bk::h2d(L.dinv, dinv.data(), sizeof(double) * L.n);
    if (l == 0) nnz0 = (double)Ah.val.size();
    nnzt += (double)Ah.val.size();
    const bool last = hp.last;
    const std::vector<int>& agg = hp.agg;
    const std::vector<int>& csub = hp.csub;
    const int nc = hp.nc;
    auto t_1 = tnow();
    if (last) {
      alloc_level_buffers(L, l > 0);
      L.fused = false;
      lv.push_back(L);
      std::vector<double> cinv;
      std::vector<int64_t> cbase;
      coarse_inverse(Ah, so, cinv, cbase);
      cch = bk::chunks_upload(nsub, so.data());
      d_inv = (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, cinv.size()));
      bk::h2d(d_inv, cinv.data(), sizeof(double) * cinv.size());
      d_invbase = (int64_t*)bk::alloc(sizeof(int64_t) * cbase.size());
      bk::h2d(d_invbase, cbase.data(), sizeof(int64_t) * cbase.size());
      break;
    }
    // ---- device: P0, A P0, P, R = P^T, A P, R A P
    const int n = Ah.n;
    int* d_agg = (int*)bk::alloc(sizeof(int) * std::max(1, n));
    bk::h2d(d_agg, agg.data(), sizeof(int) * n);
    bk::Csr P0 = bk::csr_tentative_prolongator(n, d_agg);     // rows of one entry 1.0, made on the device
    bool ok = true;
    bk::Csr P = bk::spgemm(Adev, P0, nc, &ok);
    if (!ok) { bk::csr_free(P0); bk::dfree(d_agg); bk::dfree(L.dinv); return abandon(); }
    bk::smooth_prolongator(P, d_agg, L.dinv, 4.0 / (3.0 * L.rho));
    bk::dfree(d_agg);
    bk::csr_free(P0);
    bk::Csr R = bk::transpose(P, nc, &ok);
    if (!ok) { bk::csr_free(P); bk::dfree(L.dinv); return abandon(); }
    bk::Csr AP = bk::spgemm(Adev, P, nc, &ok);
    if (!ok) { bk::csr_free(P); bk::csr_free(R); bk::dfree(L.dinv); return abandon(); }
    bk::Csr Ac = bk::spgemm(R, AP, nc, &ok);
    if (!ok) { bk::csr_free(AP); bk::csr_free(P); bk::csr_free(R); bk::dfree(L.dinv); return abandon(); }
    // next level: its matrix on the host for the aggregation / diagonal / coarsest inverse.  The download runs on a
    // helper thread with its own stream (ordered behind the product that made Ac) while this one finishes the level:
    // SpMV layouts of P, R, Ac, the column-scaled copy and the post-smoothing matrix M.
    HostCsr next = g_host_csr_pool.take(nc, (size_t)Ac.nnz);
    std::exception_ptr dl_err;
    void* parent = bk::get_stream();
    const int next_level = l + 1;
    std::thread dl([&]() {
      try {
        bk::side_stream_begin(parent, true);
        bk::csr_download(Ac, next.rowptr.data(), next.col.data(), next.val.data());
        bk::side_stream_end();
        // ... and the host part of the NEXT level (diagonal, aggregates) while the parent is still busy with this one
        hp_next = amg_level_host_part(next, csub, prm, next_level);
        have_next = true;
      } catch (...) {
        bk::side_stream_end();
        dl_err = std::current_exception();
      }
    });
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } dl_join{dl};
    bk::csr_finish(P);
    bk::csr_finish(R);
    bk::csr_finish(Ac);
    auto t_2 = tnow();
    L.P = P;
    L.R = R;
    alloc_level_buffers(L, l > 0);
    L.fused = bk::csr_fusable(L.A) && bk::csr_fusable(L.P) && !getenv("GENEO_AMG_UNFUSED");
    make_column_scaled(L);
    make_post_matrix(L, &AP, nc);       // consumes A P
    lv.push_back(L);
    dl.join();
    if (dl_err) std::rethrow_exception(dl_err);
    if (dbg)
      fprintf(stderr, "[amg/device] level n %d -> %d nnz %zu -> %zu | host diag+aggregation %.3f, device products %.3f, download %.3f s\n",
              n, nc, Ah.val.size(), next.val.size(), tsec(t_0, t_1), tsec(t_1, t_2), tsec(t_2, tnow()));
    g_host_csr_pool.give(std::move(Acur));
    Acur = std::move(next);
    Adev = Ac;
    own = true;
    so = csub;
  }
  g_host_csr_pool.give(std::move(Acur));
  opc = nnz0 > 0 ? nnzt / nnz0 : 1.0;
  if (prm.single) make_single();
  return true;
}

void AmgDevice::applyA(const bk::Csr& a, const double* X, int ldx, double* Y, int ldy, int m) {
  if (m == 1 && ldx == 1 && ldy == 1) bk::spmv(a, X, Y);
  else bk::spmm_strided(a, X, ldx, Y, ldy, m, nullptr, nullptr);
}

// Chebyshev-Jacobi smoothing of A x = b on [rho/ratio, 1.1 rho] (degree 1 = damped Jacobi)
void AmgDevice::smooth(Lvl& L, const double* B, int ldb, double* X, int ldx, int m, bool zero_guess) {
  const double lmax = 1.1 * L.rho, lmin = lmax / std::max(1.5, prm.smooth_ratio);
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  const int n = L.n;
  double* r = L.r;
  double* d = L.d;
  if (prm.smooth_degree <= 1) {  // damped Jacobi: one fused launch (+ one SpMV when the guess is non zero)
    if (!zero_guess) applyA(L.A, X, ldx, r, m, m);
    bk::jacobi_step(X, ldx, B, ldb, r, L.dinv, 1.0 / theta, n, m, zero_guess);
    return;
  }
  if (zero_guess) {
    bk::block_axpby(r, m, 1.0, B, ldb, 0.0, n, m);                         // r = b
    bk::block_rowscale(d, m, r, m, L.dinv, 1.0 / theta, 0.0, n, m);        // d = Dinv r / theta
    bk::block_axpby(X, ldx, 1.0, d, m, 0.0, n, m);                         // x = d
  } else {
    applyA(L.A, X, ldx, r, m, m);                                          // r = b - A x
    bk::block_axpby(r, m, 1.0, B, ldb, -1.0, n, m);
    bk::block_rowscale(d, m, r, m, L.dinv, 1.0 / theta, 0.0, n, m);
    bk::block_axpby(X, ldx, 1.0, d, m, 1.0, n, m);                         // x += d
  }
  double rho = 1.0 / sigma;
  for (int k = 1; k < prm.smooth_degree; ++k) {
    applyA(L.A, d, m, L.ad, m, m);                                         // A d
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    bk::cheb_update(r, L.ad, d, X, ldx, L.dinv, 2.0 * rho_new / delta, rho_new * rho, n, m);
    rho = rho_new;
  }
}

void AmgDevice::cycle(int l, const double* B, int ldb, double* X, int ldx, int m) {
  Lvl& L = lv[l];
  if (l == (int)lv.size() - 1) {
    bk::dense_sym_apply(cch, d_inv, d_invbase, B, ldb, X, ldx, m);
    return;
  }
  Lvl& C0 = lv[l + 1];
  if (prm.smooth_degree <= 1 && L.fused) {
    // damped-Jacobi V-cycle in 4 launches per level: the vector passes ride on the SpMV / SpMM epilogues
    const double w = jacobi_weight(L);
    const bk::Csr& Apre = L.Acs.n ? L.Acs : L.A;
    // contiguous single vectors: the single-precision companions apply -- only when the hierarchy was asked to use them
    // (-dls1_amg_precision single): a borrowed level-0 matrix may carry a companion for its owner's own purposes (the
    // 16-bit column offsets of the FP64 SpMV), which must not turn a "double" V-cycle into a float one
    const bool vec = (m == 1 && ldb <= 1 && ldx <= 1) && prm.single;
    // with the post-smoothing matrix x1 = w D^-1 b is never read back (EPI_POST rebuilds it from b): it is not stored
    double* X1 = L.M.n ? nullptr : X;
    if (vec && bk::csr_has_lp(Apre)) bk::spmv_fused_lp(Apre, bk::EPI_PRE, nullptr, L.r, B, X1, L.dinv, w);
    else bk::spmm_fused(Apre, bk::EPI_PRE, nullptr, 0, L.r, m, m, B, ldb, X1, ldx, L.dinv, w);   // x1 = w D^-1 b ; r1 = b - A x1
    if (vec && bk::csr_has_lp(L.R)) bk::spmv_lp(L.R, L.r, C0.b);
    else applyA(L.R, L.r, m, C0.b, m, m);                                                   // restrict
    cycle(l + 1, C0.b, m, C0.x, m, m);
    if (L.M.n) {      // x = w D^-1 (b + r1) + (P - w D^-1 A P) e
      if (vec && bk::csr_has_lp(L.M)) bk::spmv_fused_lp(L.M, bk::EPI_POST, C0.x, X, L.r, const_cast<double*>(B), L.dinv, w);
      else bk::spmm_fused(L.M, bk::EPI_POST, C0.x, m, X, ldx, m, L.r, m, const_cast<double*>(B), ldb, L.dinv, w);
      return;
    }
    if (vec && bk::csr_has_lp(L.P)) bk::spmv_fused_lp(L.P, bk::EPI_ADD, C0.x, L.d, nullptr, X, nullptr, 0.0);
    else bk::spmm_fused(L.P, bk::EPI_ADD, C0.x, m, L.d, m, m, nullptr, 0, X, ldx, nullptr, 0.0);  // t = x + P e
    if (vec && bk::csr_has_lp(L.A)) bk::spmv_fused_lp(L.A, bk::EPI_JAC, L.d, X, B, nullptr, L.dinv, w);
    else bk::spmm_fused(L.A, bk::EPI_JAC, L.d, m, X, ldx, m, B, ldb, nullptr, 0, L.dinv, w);   // x = t + w D^-1 (b - A t)
    return;
  }
  smooth(L, B, ldb, X, ldx, m, true);                                      // pre-smoothing, zero guess
  applyA(L.A, X, ldx, L.r, m, m);                                          // r = b - A x
  bk::block_axpby(L.r, m, 1.0, B, ldb, -1.0, L.n, m);
  Lvl& C = lv[l + 1];
  applyA(L.R, L.r, m, C.b, m, m);                                          // restrict
  cycle(l + 1, C.b, m, C.x, m, m);
  applyA(L.P, C.x, m, L.d, m, m);                                          // prolong + correct
  bk::block_axpby(X, ldx, 1.0, L.d, m, 1.0, L.n, m);
  smooth(L, B, ldb, X, ldx, m, false);                                     // post-smoothing
}

void AmgDevice::vcycle_from(int l, const double* B, int ldb, double* X, int ldx, int m) {
  if (m > maxm) throw std::runtime_error("AMG: block wider than the hierarchy was allocated for");
  if (l < 0 || l >= (int)lv.size()) throw std::runtime_error("AMG: no such level");
  cycle(l, B, ldb, X, ldx, m);
}

void AmgDevice::vcycle(const double* B, int ldb, double* X, int ldx, int m) {
  if (m > maxm) throw std::runtime_error("AMG: block wider than the hierarchy was allocated for");
  cycle(0, B, ldb, X, ldx, m);
}

}  // namespace geneo
