// GenEO preconditioner core: MI355X-native counterpart of /root/reference/src/geneo.cpp.
//
//   reference (PETSc/SLEPc/MUMPS objects)                     here (device-resident, FP64)
//   ---------------------------------------------------------------------------------------------
//   MATIS A, VecScatter pcScatCtx      geneo.cpp:156,1850,1881   l2e gather / R^T segmented sums + halo plan
//   A_Dir = submatrix of assembled A   geneo.cpp:1692-1705       host assembly once (or caller-supplied)
//   D = 1/multiplicity                 geneo.cpp:965-1000        d_D
//   KSP(PREONLY)+LU(MUMPS) on A_Dir    geneo.cpp:94-160,1995     batched Jacobi-PCG to dls1_rtol (one CG per subdomain)
//   EPS arpack shift-invert, GHEP      geneo.cpp:626-744         LOBPCG, Chebyshev-Jacobi preconditioner, MFMA Rayleigh-Ritz
//   inertia estimate (Sylvester)       geneo.cpp:452-560         replaced: block of cut/nev Ritz pairs, then the tau filter
//   Nicolaides / empty-Z rule          geneo.cpp:897-944,1305    same tests
//   Z (MatIS->AIJ), E = Z^T A Z, LU    geneo.cpp:355-450,1028    Z_s column-major per subdomain; E replicated, host Cholesky
//   applyQ / applyLevel1 / hybrid      geneo.cpp:1435-2098       apply_q / apply
#include "core.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <functional>
#include <future>
#include <map>
#include <numeric>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "amg.h"
#include "dense.h"

namespace geneo {

using clk = std::chrono::high_resolution_clock;
static inline double secs(clk::time_point a, clk::time_point b) {
  return std::chrono::duration<double>(b - a).count();
}

// ------------------------------------------------------------------------------------ options
std::string Options::name() const {
  std::string n = "geneo" + std::to_string(lvl2);
  if (hybrid) n += effHybrid ? "E" : "H";
  std::string l1;
  if (lvl1ASM) l1 = "ASM";
  if (lvl1RAS) l1 = "RAS";
  if (lvl1SRAS) l1 = "SRAS";
  if (lvl1ORAS) l1 = "ORAS";
  if (lvl1SRAS && lvl1ORAS) l1 = "SORAS";
  return n + l1;
}

static bool to_double(const std::string& s, double& v) {
  try {
    size_t pos = 0;
    v = std::stod(s, &pos);
    return pos > 0;
  } catch (...) {
    return false;
  }
}
static bool to_int(const std::string& s, int& v) {
  try {
    size_t pos = 0;
    v = std::stoi(s, &pos);
    return pos > 0;
  } catch (...) {
    return false;
  }
}

std::string parse_option(Options& o, const std::string& key, const std::string& value) {
  if (key == "-geneo_lvl") {  // geneo.cpp:2344-2370
    const size_t c = value.find(',');
    if (c == std::string::npos || value.find(',', c + 1) != std::string::npos) return "invalid option -geneo_lvl";
    const std::string l1 = value.substr(0, c), l2 = value.substr(c + 1);
    // a repeated -geneo_lvl replaces the earlier one (PETSc's options database keeps the last value of an option)
    o.lvl1ASM = true;
    o.lvl1RAS = o.lvl1SRAS = o.lvl1ORAS = false;
    o.hybrid = o.effHybrid = false;
    if (l1 == "ASM") o.lvl1ASM = true;
    else if (l1 == "RAS") o.lvl1RAS = true;
    else if (l1 == "SRAS") o.lvl1RAS = o.lvl1SRAS = true;
    else if (l1 == "ORAS") o.lvl1RAS = o.lvl1ORAS = true;
    else if (l1 == "SORAS") o.lvl1RAS = o.lvl1SRAS = o.lvl1ORAS = true;
    else return "invalid option -geneo_lvl, unknown " + l1;
    if (l2 == "0") { o.lvl2 = 0; }
    else if (l2 == "1") { o.lvl2 = 1; }
    else if (l2 == "H1") { o.lvl2 = 1; o.hybrid = true; }
    else if (l2 == "E1") { o.lvl2 = 1; o.hybrid = true; o.effHybrid = true; }
    else if (l2 == "2") { o.lvl2 = 2; }
    else if (l2 == "H2") { o.lvl2 = 2; o.hybrid = true; }
    else if (l2 == "E2") { o.lvl2 = 2; o.hybrid = true; o.effHybrid = true; }
    else return "invalid option -geneo_lvl, unknown " + l2;
    return "";
  }
  auto dbl = [&](double& dst) -> std::string {
    double v;
    if (!to_double(value, v)) return "invalid option " + key + ", bad " + value;
    dst = v;
    return "";
  };
  auto integer = [&](int& dst) -> std::string {
    int v;
    if (!to_int(value, v)) return "invalid option " + key + ", bad " + value;
    dst = v;
    return "";
  };
  if (key == "-geneo_optim") return dbl(o.optim);
  if (key == "-geneo_tau") return dbl(o.tau);
  if (key == "-geneo_gamma") return dbl(o.gamma);
  if (key == "-geneo_cut") return integer(o.cut);
  if (key == "-geneo_cst") { o.cst = true; return ""; }
  if (key == "-geneo_no_syl") { o.noSyl = true; return ""; }
  if (key == "-geneo_offload") { o.offload = true; return ""; }
  if (key == "-geneo_chk") {  // geneo.cpp:2466-2479 (bin / mat select PETSc viewers there; files are text here)
    if (value != "log" && value != "bin" && value != "mat") return "invalid option -geneo_chk, unknown " + value;
    o.check = true;
    return "";
  }
  if (key == "-els2_eps_tol") return dbl(o.eps_tol);
  if (key == "-els2_eps_nev") return integer(o.eps_nev);
  if (key == "-els2_eps_max_it") return integer(o.eps_max_it);
  if (key == "-els2_eps_block") return integer(o.eps_block);
  if (key == "-els2_eps_conv") {
    if (value != "sinvert" && value != "residual") return "unsupported -els2_eps_conv " + value;
    o.eps_conv = value;
    return "";
  }
  if (key == "-els2_cheb_degree") return integer(o.cheb_degree);
  if (key == "-els2_cheb_ratio") return dbl(o.cheb_ratio);
  if (key == "-els2_rr_drop") return dbl(o.rr_drop);
  if (key == "-geneo_nicolaides_zero") {
    double v;
    if (!to_double(value, v) || !(v >= 0.0)) return "invalid option -geneo_nicolaides_zero, bad " + value;
    o.nicolaides_zero = v;
    return "";
  }
  if (key == "-geneo_eig_group_rows") return integer(o.eig_group_rows);
  if (key == "-geneo_eig_mem_gb") return dbl(o.eig_mem_gb);
  if (key == "-geneo_eig_coarse_start") return integer(o.eig_coarse_start);
  if (key == "-els2_eps_seed") { int v; if (!to_int(value, v)) return "bad seed"; o.eps_seed = (uint64_t)v; return ""; }
  if (key == "-dls1_ksp_rtol") return dbl(o.dls1_rtol);
  if (key == "-dls1_ksp_max_it") return integer(o.dls1_max_it);
  if (key == "-dls1_check") return integer(o.dls1_check);
  if (key == "-dls1_pc_type") {
    if (value != "amg" && value != "jacobi") return "unsupported -dls1_pc_type " + value;
    o.dls1_pc = value;
    return "";
  }
  if (key == "-els2_pc_type") {
    if (value != "amg" && value != "cheb") return "unsupported -els2_pc_type " + value;
    o.els2_pc = value;
    return "";
  }
  if (key == "-dls1_amg_strength") return dbl(o.dls1_amg_strength);
  if (key == "-els2_amg_strength") return dbl(o.els2_amg_strength);
  if (key == "-dls1_amg_precision") {
    if (value != "single" && value != "double") return "unsupported -dls1_amg_precision " + value;
    o.dls1_amg_single = (value == "single");
    return "";
  }
  if (key == "-amg_coarse_size") return integer(o.amg_coarse_size);
  if (key == "-amg_smooth_degree") return integer(o.amg_smooth_degree);
  if (key == "-amg_smooth_ratio") return dbl(o.amg_smooth_ratio);
  if (key == "-dls1_amg_smooth_ratio") return dbl(o.dls1_amg_smooth_ratio);
  if (key == "-els2_amg_smooth_ratio") return dbl(o.els2_amg_smooth_ratio);
  if (key == "-amg_max_levels") return integer(o.amg_max_levels);
  if (key == "-ksp_type") {
    if (value != "cg" && value != "gmres") return "unsupported -ksp_type " + value;
    o.ksp_type = value;
    return "";
  }
  if (key == "-ksp_rtol") return dbl(o.ksp_rtol);
  if (key == "-ksp_atol") return dbl(o.ksp_atol);
  if (key == "-ksp_divtol") return dbl(o.ksp_dtol);
  if (key == "-ksp_max_it") return integer(o.ksp_max_it);
  if (key == "-ksp_gmres_restart") return integer(o.ksp_restart);
  if (key == "-ksp_initial_guess_nonzero") { o.ksp_guess_nonzero = (value != "0" && value != "false"); return ""; }
  return "unknown option " + key;
}

std::string validate_options(const Options& o) {  // geneo.cpp:2486-2488
  if (o.lvl2 >= 1 && o.tau <= 0.) return "GenEO preconditioner: tau must be > 0.";
  if (o.lvl2 >= 1 && o.tau >= 1.) return "GenEO preconditioner: tau must be < 1.";
  if (o.lvl2 >= 2 && o.gamma <= 1.) return "GenEO preconditioner: gamma must be > 1.";
  return "";
}

// ------------------------------------------------------------------------------------ helpers
// AMG hierarchy of a block-diagonal matrix given by its per-subdomain blocks (host set-up)
struct AmgHostResult {
  std::vector<AmgLevelHost> levels;
  std::vector<double> cinv;
  std::vector<int64_t> cbase;
  std::string err;
  double secs = 0.0;
};

struct PC::Amg1Pending {
  HostCsr mat_own;                        // block-diagonal host copy of the level-1 matrix (several subdomains) ...
  const HostCsr* matp = nullptr;          // ... or the ONE subdomain's own matrix, used where it lies (no copy)
  const HostCsr& mat() const { return *matp; }
  AmgHostResult res;
  std::future<AmgLevelHostPart> level0;   // diagonal + aggregates of the fine level, started as soon as `mat` exists
  AmgDevice* dev = nullptr;     // built / uploaded by the same thread on its side stream (handed to PC::amg1 at the join)
  double upload_secs = 0.0;
  bool on_device = false;       // sparse products on the device (host: aggregation only)
  std::thread th;
  ~Amg1Pending() {
    if (th.joinable()) th.join();
    delete dev;
  }
};

static std::string check_id(int gid, int nsub);

// Nicolaides rule (geneo.cpp:897-944): the constant vector is added when the smallest kept eigenvalue is "not zero"
// (the reference tests min >= DBL_EPSILON) and 1^T A 1 / 1^T B 1 <= FLT_EPSILON.  On an exactly singular Neumann matrix
// (--inpEps 0) the computed zero eigenvalue is a rounding error of either sign and any size around 1e-16 -- LAPACK returns
// +2e-15 on two of the four floating subdomains of a 20^3 grid, LOBPCG on the GPU +4e-16 on one --, and whenever it lands
// above DBL_EPSILON the reference's test adds the kernel vector a SECOND time: a rank-deficient Z and a singular E.
// An eigenvalue below 100 DBL_EPSILON is therefore taken as the zero it is by default; outside that window (every case
// the reference's own tests exercise) the rule is the reference's.  -geneo_nicolaides_zero <x> sets the window to
// x DBL_EPSILON: 1 is the reference's literal test (geneo.cpp:898-899).

int PC::fail(const std::string& msg) {
  last_error = msg;
  return 1;
}
PC::PC() = default;
PC::~PC() { free_all(); }

void PC::free_all() {
  pend1.reset();
  for (auto& kv : cg_graphs) bk::graph_destroy(kv.second);
  cg_graphs.clear();
  cg_graph_failed = false;
  cg_long_len = 0;
  delete amg1;
  delete amgN;
  amg1 = amgN = nullptr;
  bk::csr_free(dirL);
  bk::csr_free(neuE);
  bk::csr_free(neuL);
  if (ch.start) bk::chunks_free(ch);
  void* ptrs[] = {d_l2e, d_rt_ptr, d_rt_idx, d_send_idx, d_rv_ptr, d_rv_idx, d_rv_tgt, d_D, d_dinv1, d_dinvN, d_xe,
                  d_ye, d_xL, d_wL, d_cg_r, d_cg_z, d_cg_p, d_cg_q, d_cg_sc, d_t1, d_t2, d_t3, d_x0, d_scal,
                  d_rvtmp, d_Z, d_zbase, d_ksub, d_zoff, d_subgid, d_yE, d_EL, d_ELT};
  for (void* p : ptrs) bk::dfree(p);
  d_l2e = d_rt_ptr = d_rt_idx = d_send_idx = d_rv_ptr = d_rv_idx = d_rv_tgt = nullptr;
  d_D = d_dinv1 = d_dinvN = d_xe = d_ye = d_xL = d_wL = nullptr;
  d_cg_r = d_cg_z = d_cg_p = d_cg_q = d_cg_sc = d_t1 = d_t2 = d_t3 = d_x0 = d_scal = d_rvtmp = nullptr;
  d_Z = nullptr; d_zbase = nullptr; d_ksub = d_zoff = d_subgid = nullptr; d_yE = nullptr; d_EL = d_ELT = nullptr;
  is_setup = false;
}

static void copy_csr(HostCsr& dst, int n, const int* rp, const int* col, const double* val) {
  dst.n = n;
  dst.rowptr.assign(rp, rp + n + 1);
  dst.col.assign(col, col + rp[n]);
  dst.val.assign(val, val + rp[n]);
}

int PC::add_subdomain(int gid, int n, const int* l2g, const int* mult, const int* nrp, const int* ncol,
                      const double* nval, const int* drp, const int* dcol, const double* dval) {
  if (n < 0 || !l2g || !mult || !nrp || !ncol || !nval) return fail("GenEO preconditioner: bad subdomain arguments");
  Sub s;
  s.gid = gid;
  s.l2g.assign(l2g, l2g + n);
  s.mult.assign(mult, mult + n);
  for (int i = 1; i < n; ++i)
    if (s.l2g[i] <= s.l2g[i - 1]) return fail("GenEO preconditioner: local-to-global map must be ascending");
  for (int i = 0; i < n; ++i)
    if (s.mult[i] < 1) return fail("GenEO preconditioner bad DOF multiplicity");
  copy_csr(s.a_neu, n, nrp, ncol, nval);
  if (drp) copy_csr(s.a_dir, n, drp, dcol, dval);
  subs.push_back(std::move(s));
  return 0;
}

int PC::set_intersect(int gid, int nb, const int* nonempty) {
  if (nb < 0 || (nb > 0 && !nonempty)) return fail("GenEO preconditioner: bad intersection flags");
  if (gid < 0 && !subs.empty()) {   // the subdomain added last (initGenEOPC: the caller does not know the id)
    subs.back().intersect.assign(nonempty, nonempty + nb);
    return 0;
  }
  for (auto& s : subs)
    if (s.gid == gid) {
      s.intersect.assign(nonempty, nonempty + nb);
      return 0;
    }
  return fail("GenEO preconditioner: intersection flags for an unknown subdomain");
}

// ------------------------------------------------------------------------------------ layout
// f(r0, r1) over contiguous ranges of [0, n) on up to 16 host threads (serial below 100 000 items).  With one big
// subdomain per rank (the N > 1 layout of bench.py) the per-subdomain loops of the set-up are single loops over
// millions of rows.
static void parallel_ranges(int64_t n, const std::function<void(int64_t, int64_t)>& f) {
  int nth = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (n < 100000) nth = 1;
  if (nth == 1) {
    f(0, n);
    return;
  }
  std::vector<std::thread> th;
  std::exception_ptr err;
  std::mutex mu;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      try {
        f(n * t / nth, n * (t + 1) / nth);
      } catch (...) {
        std::lock_guard<std::mutex> lk(mu);
        if (!err) err = std::current_exception();
      }
    });
  for (auto& x : th) x.join();
  if (err) std::rethrow_exception(err);
}

int PC::build_layout() {
  const int ns = (int)subs.size();
  if (size == 1 && owned.empty()) {
    owned.resize(N);
    std::iota(owned.begin(), owned.end(), 0);
  }
  const int nown = n_owned();
  nH = (int)halo_gid.size();
  nE = nown + nH;
  suboff.assign(ns + 1, 0);
  for (int s = 0; s < ns; ++s) suboff[s + 1] = suboff[s] + (int)subs[s].l2g.size();
  nL = suboff[ns];
  std::vector<std::pair<int, int>> hmap(nH);
  for (int i = 0; i < nH; ++i) hmap[i] = {halo_gid[i], i};
  std::sort(hmap.begin(), hmap.end());
  const bool ident = (nown == N);
  std::vector<int> l2e(nL);
  {
    // one pass over the local space, whatever the subdomains (a parallel loop per subdomain spawned 8 x 16 threads)
    std::atomic<int> bad{0};      // 1: global index out of range, 2: neither owned nor in the halo plan
    parallel_ranges((int64_t)nL, [&](int64_t i0, int64_t i1) {
      int s = (int)(std::upper_bound(suboff.begin(), suboff.end(), (int)i0) - suboff.begin()) - 1;
      for (int64_t i = i0; i < i1;) {
        while (s + 1 < ns && i >= suboff[s + 1]) ++s;
        const std::vector<int>& l2g = subs[s].l2g;
        const int64_t e1 = std::min<int64_t>(i1, suboff[s + 1]);
        for (; i < e1; ++i) {
          const int g = l2g[i - suboff[s]];
          int e = -1;
          if (g < 0 || g >= N) { bad = 1; return; }
          if (ident) e = g;
          else {
            auto it = std::lower_bound(owned.begin(), owned.end(), g);
            if (it != owned.end() && *it == g) e = (int)(it - owned.begin());
            else {
              auto h = std::lower_bound(hmap.begin(), hmap.end(), std::make_pair(g, -1));
              if (h == hmap.end() || h->first != g) { bad = 2; return; }
              e = nown + h->second;
            }
          }
          l2e[i] = e;
        }
      }
    });
    if (bad == 1) return fail("GenEO preconditioner: global index out of range");
    if (bad == 2) return fail("GenEO preconditioner: DOF neither owned nor in the halo plan");
  }
  // R^T as CSR over the ext space (entries in ascending local index => fixed summation order).  R is a matrix of one
  // entry per row (local j -> column l2e[j]): its transpose is made by the device transposition of the multigrid set-up
  // (counts, scan, scatter, rows sorted by source row) -- the three serial host loops over the local space it replaces
  // were 20 ms of the set-up of a 6.5 M-row subdomain.  Host fallback when a DOF is shared by more subdomains than a
  // transposed row holds.
  d_l2e = (int*)bk::alloc(sizeof(int) * std::max(1, nL));
  bk::h2d(d_l2e, l2e.data(), sizeof(int) * nL);
  bool rt_done = false;
  if (nL > 0 && !getenv("GENEO_RT_HOST")) {
    bk::Csr Rm = bk::csr_tentative_prolongator(nL, d_l2e);
    bool ok = false;
    bk::Csr Rt = bk::transpose(Rm, nE, &ok);
    bk::csr_free(Rm);
    if (ok) {
      d_rt_ptr = Rt.rowptr;
      d_rt_idx = Rt.col;
      bk::dfree(Rt.val);
      rt_done = true;
    }
  }
  if (!rt_done) {
    std::vector<int> rt_ptr(nE + 1, 0), rt_idx(nL);
    for (int j = 0; j < nL; ++j) rt_ptr[l2e[j] + 1]++;
    for (int e = 0; e < nE; ++e) rt_ptr[e + 1] += rt_ptr[e];
    {
      std::vector<int> fill(rt_ptr.begin(), rt_ptr.end() - 1);
      for (int j = 0; j < nL; ++j) rt_idx[fill[l2e[j]]++] = j;
    }
    d_rt_ptr = (int*)bk::alloc(sizeof(int) * (nE + 1));
    d_rt_idx = (int*)bk::alloc(sizeof(int) * std::max(1, nL));
    bk::h2d(d_rt_ptr, rt_ptr.data(), sizeof(int) * (nE + 1));
    bk::h2d(d_rt_idx, rt_idx.data(), sizeof(int) * nL);
  }
  // halo plan: forward pack list and reverse-add lists (per owned DOF, ascending recv position)
  const int nsend = (int)send_idx.size();
  if (size > 1) {
    if ((int)recv_counts.size() != size || (int)send_counts.size() != size) return fail("GenEO: bad halo plan");
    if (std::accumulate(recv_counts.begin(), recv_counts.end(), 0) != nH) return fail("GenEO: bad halo recv counts");
    if (std::accumulate(send_counts.begin(), send_counts.end(), 0) != nsend) return fail("GenEO: bad halo send counts");
    if (!cb_exchange || !cb_allreduce || !comm_send || !comm_recv || !comm_red) return fail("GenEO: communicator not set");
    d_send_idx = (int*)bk::alloc(sizeof(int) * std::max(1, nsend));
    bk::h2d(d_send_idx, send_idx.data(), sizeof(int) * nsend);
    std::vector<int> rv_ptr(nown + 1, 0), rv_idx(nsend);
    for (int k = 0; k < nsend; ++k) {
      if (send_idx[k] < 0 || send_idx[k] >= nown) return fail("GenEO: bad halo send index");
      rv_ptr[send_idx[k] + 1]++;
    }
    for (int e = 0; e < nown; ++e) rv_ptr[e + 1] += rv_ptr[e];
    std::vector<int> fill(rv_ptr.begin(), rv_ptr.end() - 1);
    for (int k = 0; k < nsend; ++k) rv_idx[fill[send_idx[k]]++] = k;
    d_rv_ptr = (int*)bk::alloc(sizeof(int) * (nown + 1));
    d_rv_idx = (int*)bk::alloc(sizeof(int) * std::max(1, nsend));
    bk::h2d(d_rv_ptr, rv_ptr.data(), sizeof(int) * (nown + 1));
    bk::h2d(d_rv_idx, rv_idx.data(), sizeof(int) * nsend);
    n_rv = nsend;
  }
  ch = bk::chunks_upload(ns, suboff.data());
  std::vector<int> gids(std::max(1, ns));
  for (int s = 0; s < ns; ++s) gids[s] = subs[s].gid;
  d_subgid = (int*)bk::alloc(sizeof(int) * std::max(1, ns));
  bk::h2d(d_subgid, gids.data(), sizeof(int) * ns);
  // work vectors
  auto dv = [](size_t n) { return (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, n)); };
  d_xe = dv(nE); d_ye = dv(nE); d_xL = dv(nL); d_wL = dv(nL);
  d_cg_r = dv(nL); d_cg_z = dv(nL); d_cg_p = dv(nL); d_cg_q = dv(nL); d_cg_sc = dv((size_t)8 * std::max(1, ns));
  d_t1 = dv(nown); d_t2 = dv(nown); d_t3 = dv(nown); d_x0 = dv(nown); d_scal = dv(16);
  return 0;
}

// A_Dir,i = R_i (sum_j R_j^T A_Neu,j R_j) R_i^T  -- MatConvert + MatCreateSubMatrices, geneo.cpp:1692-1705
int PC::ensure_dirichlet() {
  bool missing = false;
  for (auto& s : subs)
    if (s.a_dir.empty()) missing = true;
  if (!missing) return 0;
  if (size > 1)
    return fail("GenEO preconditioner without dirichlet matrix (pass pcADirLoc when subdomains span several ranks)");
  // assemble the global matrix (COO -> sorted CSR), then extract
  size_t tot = 0;
  for (auto& s : subs) tot += s.a_neu.val.size();
  struct Ent { int64_t key; double v; };
  std::vector<Ent> coo;
  coo.reserve(tot);
  for (auto& s : subs)
    for (int i = 0; i < s.a_neu.n; ++i)
      for (int k = s.a_neu.rowptr[i]; k < s.a_neu.rowptr[i + 1]; ++k)
        coo.push_back({(int64_t)s.l2g[i] * N + s.l2g[s.a_neu.col[k]], s.a_neu.val[k]});
  std::stable_sort(coo.begin(), coo.end(), [](const Ent& a, const Ent& b) { return a.key < b.key; });
  std::vector<int64_t> keys;
  std::vector<double> vals;
  keys.reserve(coo.size());
  vals.reserve(coo.size());
  for (size_t i = 0; i < coo.size();) {
    size_t j = i;
    double v = 0.0;
    while (j < coo.size() && coo[j].key == coo[i].key) v += coo[j++].v;
    keys.push_back(coo[i].key);
    vals.push_back(v);
    i = j;
  }
  std::vector<int64_t> rowstart(N + 1, 0);
  for (int64_t k : keys) rowstart[k / N + 1]++;
  for (int i = 0; i < N; ++i) rowstart[i + 1] += rowstart[i];
  std::vector<int> g2l(N, -1);
  for (auto& s : subs) {
    if (!s.a_dir.empty()) continue;
    const int n = (int)s.l2g.size();
    for (int i = 0; i < n; ++i) g2l[s.l2g[i]] = i;
    s.a_dir.n = n;
    s.a_dir.rowptr.assign(n + 1, 0);
    s.a_dir.col.clear();
    s.a_dir.val.clear();
    for (int i = 0; i < n; ++i) {
      const int g = s.l2g[i];
      for (int64_t k = rowstart[g]; k < rowstart[g + 1]; ++k) {
        const int gc = (int)(keys[k] % N);
        if (g2l[gc] >= 0) {
          s.a_dir.col.push_back(g2l[gc]);
          s.a_dir.val.push_back(vals[k]);
        }
      }
      s.a_dir.rowptr[i + 1] = (int)s.a_dir.col.size();
    }
    for (int i = 0; i < n; ++i) g2l[s.l2g[i]] = -1;
  }
  return 0;
}

// createRobinMatrix, geneo.cpp:1613-1670: A_Rob = A_Dir + optim * A_Neu[border, border]
void PC::make_robin(Sub& s, HostCsr& out) const {
  out = s.a_dir;
  if (std::fabs(opt.optim) <= DBL_EPSILON) return;
  const int n = (int)s.l2g.size();
  std::vector<std::map<int, double>> rows(n);
  for (int i = 0; i < n; ++i)
    for (int k = s.a_dir.rowptr[i]; k < s.a_dir.rowptr[i + 1]; ++k) rows[i][s.a_dir.col[k]] += s.a_dir.val[k];
  for (int i = 0; i < n; ++i) {
    if (s.mult[i] <= 1) continue;
    for (int k = s.a_neu.rowptr[i]; k < s.a_neu.rowptr[i + 1]; ++k) {
      const int c = s.a_neu.col[k];
      if (s.mult[c] > 1) rows[i][c] += opt.optim * s.a_neu.val[k];
    }
  }
  out.rowptr.assign(n + 1, 0);
  out.col.clear();
  out.val.clear();
  for (int i = 0; i < n; ++i) {
    for (auto& kv : rows[i]) {
      out.col.push_back(kv.first);
      out.val.push_back(kv.second);
    }
    out.rowptr[i + 1] = (int)out.col.size();
  }
}

// `b` may come in with the arrays of an earlier set-up: they are reused (same sizes: no reallocation, no zero fill, no
// page faults -- a third of the 0.035 s this copy costs at 126^3)
static void make_blockdiag(const std::vector<const HostCsr*>& mats, const std::vector<int>& suboff,
                           const int* colmap /*nullable: L -> ext*/, HostCsr& b) {
  const int ns = (int)mats.size();
  b.n = suboff[ns];
  std::vector<size_t> nzoff(ns + 1, 0);
  for (int s = 0; s < ns; ++s) nzoff[s + 1] = nzoff[s] + mats[s]->val.size();
  const bool big = nzoff[ns] > 1000000;
  {   // the two large arrays are sized (and their pages first touched) side by side
    std::thread tv;
    if (big) tv = std::thread([&]() { b.val.resize(nzoff[ns]); });
    else b.val.resize(nzoff[ns]);
    b.col.resize(nzoff[ns]);
    b.rowptr.resize((size_t)b.n + 1);
    if (tv.joinable()) tv.join();
  }
  // Row ranges of about equal size over all the matrices: with ONE subdomain per rank (the N > 1 layout of bench.py) a
  // thread per subdomain is a single thread copying 44 M entries (184^3: 0.3 s of the set-up).
  struct Task { int s, r0, r1; };
  std::vector<Task> tasks;
  const int nth = big ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
  const size_t per = std::max<size_t>(200000, nzoff[ns] / (size_t)nth + 1);
  for (int s = 0; s < ns; ++s) {
    const HostCsr& m = *mats[s];
    int r0 = 0;
    while (r0 < m.n) {
      int r1 = r0;
      const size_t lim = (size_t)m.rowptr[r0] + per;
      r1 = (int)(std::upper_bound(m.rowptr.begin() + r0, m.rowptr.begin() + m.n + 1, (int)std::min<size_t>(lim, 0x7fffffff)) -
                 m.rowptr.begin()) - 1;
      r1 = std::max(r0 + 1, std::min(r1, m.n));
      tasks.push_back({s, r0, r1});
      r0 = r1;
    }
  }
  b.rowptr[0] = 0;
  auto fill = [&](const Task& t) {
    const HostCsr& m = *mats[t.s];
    const int off = suboff[t.s];
    for (int i = t.r0; i < t.r1; ++i) {
      size_t pos = nzoff[t.s] + (size_t)m.rowptr[i];
      for (int k = m.rowptr[i]; k < m.rowptr[i + 1]; ++k, ++pos) {
        const int cl = off + m.col[k];
        b.col[pos] = colmap ? colmap[cl] : cl;
        b.val[pos] = m.val[k];
      }
      b.rowptr[off + i + 1] = (int)pos;
    }
  };
  if (nth > 1 && tasks.size() > 1) {
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t)
      th.emplace_back([&, t]() {
        for (size_t k = t; k < tasks.size(); k += nth) fill(tasks[k]);
      });
    for (auto& x : th) x.join();
  } else {
    for (const Task& t : tasks) fill(t);
  }
}
static bk::Csr upload_host(const HostCsr& b) { return bk::csr_upload(b.n, b.rowptr.data(), b.col.data(), b.val.data()); }
static bk::Csr upload_blockdiag(const std::vector<const HostCsr*>& mats, const std::vector<int>& suboff,
                                const int* colmap /*nullable: L -> ext*/) {
  HostCsr b;
  make_blockdiag(mats, suboff, colmap, b);
  return upload_host(b);
}

static AmgParams amg_params(const Options& opt) {
  AmgParams ap;
  ap.coarse_size = opt.amg_coarse_size;
  ap.smooth_degree = opt.amg_smooth_degree;
  ap.smooth_ratio = opt.amg_smooth_ratio;
  ap.max_levels = opt.amg_max_levels;
  return ap;
}

// Joins the host set-up of the level-1 hierarchy (started in setup) and uploads it.
int PC::finish_amg1() {
  if (!pend1) return 0;
  auto t0 = clk::now();
  if (pend1->th.joinable()) pend1->th.join();
  const double waited = secs(t0, clk::now());
  std::unique_ptr<Amg1Pending> p(pend1.release());
  if (!p->res.err.empty()) return fail(p->res.err);
  amg1 = p->dev;             // built and uploaded by the thread, on its side stream, while this one ran the eigensolve
  p->dev = nullptr;
  info.amg_levels = amg1->nlevels();
  info.amg_operator_complexity = amg1->operator_complexity();
  if (getenv("GENEO_DEBUG")) fprintf(stderr, "[amg] level-1 hierarchy: %d matrices with a single-precision companion\n", amg1->lp_matrices());
  const double tot = secs(t0, clk::now());
  info.amgSetupTime += tot;
  info.lvl1SetupMinvTimeLoc += tot;
  if (getenv("GENEO_DEBUG"))
    fprintf(stderr, "[amg] level-1 hierarchy on its own thread and stream (%s): %.3f s%s; waited %.3f s for it\n",
            p->on_device ? "device products, host aggregation" : "host products", p->res.secs,
            p->on_device ? "" : (" + upload " + std::to_string(p->upload_secs) + " s").c_str(), waited);
  return 0;
}

// ------------------------------------------------------------------------------------ setup
int PC::setup(const double* b_dev) {
  // Memory-bounded set-up (several groups of eigensolves): this PC's own device preparation -- 8.8 GB of fine matrices
  // through the pinned staging buffers, their sliced layouts and companions, the diagonals, the start of the level-1
  // hierarchy: 0.4 s at 368^3 -- runs on a helper thread and side stream WHILE this thread runs the grouped eigensolves,
  // which need none of it.  The helper first makes the host copies it needs of the subdomain matrices (block-diagonal
  // forms, partition of unity) and then releases the subdomains (`subs_released`) for eigen_grouped, which moves their
  // matrices in and out of its temporary PCs.  GENEO_SETUP_NO_OVERLAP=1: one after the other.
  bool overlap = false;
  if (!eig_only && subs.size() >= 2 && opt.lvl2 == 1 && !opt.check && opt.els2_pc == "amg" && !getenv("GENEO_SETUP_NO_OVERLAP")) {
    int nmax = 0;
    for (auto& s : subs) nmax = std::max(nmax, (int)s.l2g.size());
    overlap = nmax > 192 && validate_options(opt).empty() && N > 0 && plan_eig_groups().size() > 2;
  }
  if (!overlap) {
    if (int rc = setup_prepare()) return rc;
    return setup_finish(b_dev);
  }
  const auto t0 = clk::now();
  std::promise<bool> released;
  std::future<bool> subs_free = released.get_future();
  subs_released = &released;
  int rc_prep = 1;
  std::string err_prep;
  std::thread helper([&]() {
    try {
      bk::side_stream_begin();
      rc_prep = setup_prepare();
      if (rc_prep) err_prep = last_error;
      bk::side_stream_end();
    } catch (std::exception& e) {
      bk::side_stream_end();
      err_prep = e.what();
    }
    if (subs_released) {      // an early return: nothing was released
      subs_released = nullptr;
      released.set_value(false);
    }
  });
  int rc_eig = 0;
  std::string err_eig;
  if (subs_free.get()) {      // host copies made: the subdomains are this thread's now
    try {
      rc_eig = setup_level2_eigen();
      if (rc_eig) err_eig = last_error;
    } catch (std::exception& e) {
      rc_eig = 1;
      err_eig = e.what();
    }
    eig_early = (rc_eig == 0);
  }
  helper.join();
  if (rc_prep) { eig_early = false; is_setup = false; return fail(err_prep.empty() ? "GenEO preconditioner: set-up failed" : err_prep); }
  if (rc_eig) { eig_early = false; is_setup = false; return fail(err_eig); }
  const int rc = setup_finish(b_dev);
  eig_early = false;
  info.setupTime = secs(t0, clk::now()) - release_secs;     // wall clock of both threads, the release of the previous set-up excluded
  return rc;
}

// D = 1 / mult on the host (geneo.cpp:965-1000), into h_Dscratch: kept from one set-up to the next (no page faults, no
// zero fill); one pass over the local space, whatever the subdomains
void PC::fill_partition_of_unity() {
  const int ns = (int)subs.size();
  std::vector<double>& D = h_Dscratch;
  if ((int)D.size() != std::max(1, nL)) D.resize(std::max(1, nL));
  parallel_ranges((int64_t)nL, [&](int64_t i0, int64_t i1) {
    int s = (int)(std::upper_bound(suboff.begin(), suboff.end(), (int)i0) - suboff.begin()) - 1;
    for (int64_t i = i0; i < i1;) {
      while (s + 1 < ns && i >= suboff[s + 1]) ++s;
      const auto& mult = subs[s].mult;
      const int64_t e = std::min<int64_t>(i1, suboff[s + 1]);
      for (; i < e; ++i) D[i] = 1.0 / (double)mult[i - suboff[s]];
    }
  });
}

// First half of the set-up: layout, matrices on the device, diagonals, the A_Neu hierarchy (and the start of the level-1
// one) -- everything the eigensolve waits for.  eigen_grouped runs it for the NEXT group on a helper thread and a side
// stream while the main stream iterates on the current one.
int PC::setup_prepare() {
  // a second set-up of the same PC (or a retry after a failed one) first releases everything the previous one
  // allocated; the clock of setupTime starts after that release
  const auto t_rel = clk::now();
  free_all();
  release_secs = secs(t_rel, clk::now());
  if (getenv("GENEO_DEBUG")) fprintf(stderr, "[setup] release of the previous set-up (outside setupTime) %.4f s\n", release_secs);
  info = Info();
  if (!eig_only) (void)amg_null_pivots_take();     // (a group of eigen_grouped counts into its owner's total)
  auto t0 = clk::now();
  std::string err = validate_options(opt);
  if (!err.empty()) return fail(err);
  if (opt.lvl2 == 2 && !opt.lvl1ORAS)
    return fail("GenEO-2 needs the Robin matrix: use -geneo_lvl ORAS,2 or SORAS,2 (geneo.cpp:1283 takes pcARobLoc)");
  if (N <= 0) return fail("GenEO preconditioner: empty problem");
  if (int rc = build_layout()) return rc;
  if (getenv("GENEO_DEBUG")) fprintf(stderr, "[setup] %-28s %.3f s\n", "layout (maps, R^T, work vectors)", secs(t0, clk::now()));
  if (int rc = ensure_dirichlet()) return rc;
  const int ns = (int)subs.size();
  for (auto& s : subs) {
    if (s.a_neu.n != (int)s.l2g.size() || s.a_dir.n != (int)s.l2g.size())
      return fail("GenEO preconditioner: local matrix size mismatch");
  }
  // level-1 matrices (Dirichlet, or Robin for ORAS: geneo.cpp:137-144)
  std::vector<HostCsr> rob(ns);
  std::vector<const HostCsr*> neu(ns), lvl1(ns);
  for (int s = 0; s < ns; ++s) {
    neu[s] = &subs[s].a_neu;
    if (opt.lvl1ORAS) {
      make_robin(subs[s], rob[s]);
      lvl1[s] = &rob[s];
    } else {
      lvl1[s] = &subs[s].a_dir;
    }
  }
  auto t1 = clk::now();
  // the level-1 block-diagonal matrix is assembled on its own thread while this one assembles and uploads A_Neu
  // With ONE subdomain on this rank (the N > 1 layout of bench.py: a 6.5 M-row block) the block-diagonal matrices ARE the
  // subdomain's: they are used where they lie instead of being copied (0.085 s per matrix at that size).
  const bool single_block = (ns == 1);
  HostCsr& h_dirL_own = host_dir_cache;      // kept between set-ups of this PC (released in PCDestroy / when sizes change)
  HostCsr& h_neuL_own = host_neu_cache;
  std::exception_ptr dir_err;      // a bad_alloc on the thread must come back as an error code, not std::terminate
  // The level-1 matrix is assembled AND uploaded (with its 16-bit column offsets / float values: the FP64 SpMV of the
  // local solves then reads 10 bytes per entry instead of 12, the level-1 hierarchy borrows the companion) by its own
  // thread on a side stream, next to the assembly + upload of A_Neu on this one: two staging sets, two host memcpy
  // streams into pinned memory.
  // memory-bounded set-up: the eigensolve of this rank's subdomains in consecutive groups (eigen_grouped), each with its
  // own A_Neu hierarchy -- this PC then builds none
  eig_groups = plan_eig_groups();
  const bool grouped = eig_groups.size() > 2;
  info.eig_groups = (int)eig_groups.size() - 1;
  // overlapped with the grouped eigensolves (PC::setup): both host copies and the partition of unity first, then the
  // subdomains are released to the thread that runs eigen_grouped; nothing below reads `subs` any more
  bool host_copies_done = false;
  if (subs_released && grouped && !single_block) {
    std::exception_ptr herr;
    std::thread hc([&]() {
      try { make_blockdiag(lvl1, suboff, nullptr, h_dirL_own); } catch (...) { herr = std::current_exception(); }
    });
    try { make_blockdiag(neu, suboff, nullptr, h_neuL_own); } catch (...) { if (!herr) herr = std::current_exception(); }
    hc.join();
    if (herr) return fail("GenEO preconditioner: host copies of the subdomain matrices failed (out of memory?)");
    fill_partition_of_unity();
    host_copies_done = true;
    std::promise<bool>* pr = subs_released;
    subs_released = nullptr;
    pr->set_value(true);
  }
  std::thread dir_thread([&]() {
    try {
      if (!single_block && !host_copies_done) make_blockdiag(lvl1, suboff, nullptr, h_dirL_own);
      bk::side_stream_begin(bk::get_stream(), true);
      dirL = upload_host(single_block ? *lvl1[0] : h_dirL_own);
      dirL.fine = true;
      bk::csr_make_lp(dirL);
      bk::side_stream_end();
    } catch (...) {
      bk::side_stream_end();
      dir_err = std::current_exception();
    }
  });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } dir_joiner{dir_thread};
  if (!single_block && !host_copies_done) make_blockdiag(neu, suboff, nullptr, h_neuL_own);
  const HostCsr& h_neuL = single_block ? *neu[0] : h_neuL_own;
  const bool want1 = (opt.dls1_pc == "amg") && !eig_only;
  const bool wantN = (opt.lvl2 && opt.els2_pc == "amg") && !grouped;
  const AmgParams ap = amg_params(opt);
  AmgParams ap1h = ap;                      // the level-1 hierarchy (local solves): its own aggregation strength
  ap1h.strength = opt.dls1_amg_strength;
  if (opt.dls1_amg_smooth_ratio > 0.0) ap1h.smooth_ratio = opt.dls1_amg_smooth_ratio;
  AmgParams apN = ap;                       // the A_Neu hierarchy (LOBPCG preconditioner)
  apN.strength = opt.els2_amg_strength;
  if (opt.els2_amg_smooth_ratio > 0.0) apN.smooth_ratio = opt.els2_amg_smooth_ratio;
  // The host part of the fine level of both hierarchies (Jacobi diagonal, Gershgorin bound, aggregates) needs nothing
  // but the host matrices: it runs on its own threads from here on, behind the uploads and the diagonals (with ONE
  // 6.5 M-row subdomain per GPU it is 0.08-0.1 s per hierarchy of a single thread).
  const bool early0 = !getenv("GENEO_AMG_HOST") && !getenv("GENEO_AMG_LATE_LEVEL0");
  std::future<AmgLevelHostPart> pre_neu;
  if (wantN && early0)
    pre_neu = std::async(std::launch::async, [&h_neuL, this, apN]() { return amg_level_host_part(h_neuL, suboff, apN, 0); });
  auto tdbg = clk::now();
  auto lap = [&](const char* what) {
    if (!getenv("GENEO_DEBUG")) return;
    bk::sync();
    fprintf(stderr, "[setup] %-28s %.3f s\n", what, secs(tdbg, clk::now()));
    tdbg = clk::now();
  };
  if (getenv("GENEO_DEBUG")) fprintf(stderr, "[setup] %-28s %.3f s\n", "layout + Dirichlet + A_Neu blockdiag", secs(t0, tdbg));
  neuL = upload_host(h_neuL);
  neuL.fine = true;
  lap("upload A_Neu");
  // same matrix with ext-space columns (l2e o col) for the MATIS MatMult, so that the gather R x is fused into the
  // SpMV; own copy (the sliced layout embeds the columns), made on the device from the one just uploaded
  if (!eig_only) neuE = bk::csr_remap_columns(neuL, d_l2e);
  lap("ext-space copy");
  dir_thread.join();
  if (dir_err) {
    try {
      std::rethrow_exception(dir_err);
    } catch (std::exception& e) {
      return fail(std::string("GenEO preconditioner: level-1 matrix assembly failed: ") + e.what());
    }
  }
  lap("A_Dir blockdiag (joined)");
  pend1.reset(want1 ? new Amg1Pending() : nullptr);
  if (want1) {
    if (single_block && opt.lvl1ORAS) {
      // the Robin matrix is a local of this function: the hierarchy thread reads it until it is joined (finish_amg1, or
      // ~Amg1Pending on an error path), so it moves into the pending object and lives exactly as long as the thread
      pend1->mat_own = std::move(rob[0]);
      pend1->matp = lvl1[0] = &pend1->mat_own;
    } else if (single_block) {
      pend1->matp = lvl1[0];                       // subs[0].a_dir: a member of this PC, outlives the thread
    } else {
      pend1->matp = &h_dirL_own;                   // the PC's own copy: outlives the thread
    }
    if (early0 && !getenv("GENEO_AMG1_HOST")) {
      Amg1Pending* pp = pend1.get();
      const std::vector<int> so = suboff;
      pend1->level0 = std::async(std::launch::async, [pp, so, ap1h]() { return amg_level_host_part(pp->mat(), so, ap1h, 0); });
    }
  }
  lap("A_Dir uploaded by its thread (joined)");
  // partition of unity (geneo.cpp:965-1000) and Jacobi diagonals
  {
    if (!host_copies_done) fill_partition_of_unity();
    std::vector<double>& D = h_Dscratch;
    d_D = (double*)bk::alloc(sizeof(double) * std::max(1, nL));
    bk::h2d(d_D, D.data(), sizeof(double) * nL);
    d_dinv1 = (double*)bk::alloc(sizeof(double) * std::max(1, nL));
    d_dinvN = (double*)bk::alloc(sizeof(double) * std::max(1, nL));
    // Jacobi diagonals inverted and checked on the device (round 2 took both diagonals to the host and back)
    bk::csr_diag(dirL, d_dinv1);
    if (bk::recip_positive(d_dinv1, nL)) return fail("GenEO preconditioner: non-positive diagonal in the local Dirichlet matrix");
    bk::csr_diag(neuL, d_dinvN);
    // Gershgorin bounds of the Jacobi-scaled matrices: only the Chebyshev fallbacks of the eigensolves read them
    const bool need_lmax = opt.lvl2 && (opt.els2_pc != "amg" || opt.check);
    const bool need_lmax1 = opt.lvl2 == 2 && (opt.els2_pc != "amg" || opt.dls1_pc != "amg");
    if (need_lmax) {
      std::vector<double> dg(std::max(1, nL));
      bk::d2h(dg.data(), d_dinvN, sizeof(double) * nL);
      double lmax = 0.0;
      for (int s = 0; s < ns; ++s) {
        const HostCsr& m = subs[s].a_neu;
        std::mutex mu;
        parallel_ranges(m.n, [&](int64_t i0, int64_t i1) {
          double lm = 0.0;
          for (int64_t i = i0; i < i1; ++i) {
            double row = 0.0;
            for (int k = m.rowptr[i]; k < m.rowptr[i + 1]; ++k) row += std::fabs(m.val[k]);
            const double d = dg[suboff[s] + i];
            if (d > 0.0) lm = std::max(lm, row / d);
          }
          std::lock_guard<std::mutex> lk(mu);
          lmax = std::max(lmax, lm);
        });
      }
      cheb_lmax = lmax > 0 ? lmax : 2.0;
    }
    if (need_lmax1) {  // same bound for the level-1 (Robin) matrix: Chebyshev fallback of the gamma eigenproblem
      std::vector<double> d1(std::max(1, nL));
      bk::d2h(d1.data(), d_dinv1, sizeof(double) * nL);
      double l1 = 0.0;
      for (int s = 0; s < ns; ++s) {
        const HostCsr& m = *lvl1[s];
        for (int i = 0; i < m.n; ++i) {
          double row = 0.0;
          for (int k = m.rowptr[i]; k < m.rowptr[i + 1]; ++k) row += std::fabs(m.val[k]);
          l1 = std::max(l1, row * d1[suboff[s] + i]);
        }
      }
      cheb_lmax1 = l1 > 0 ? l1 : 2.0;
    }
    if (bk::recip_positive(d_dinvN, nL)) return fail("GenEO preconditioner: non-positive diagonal in the local Neumann matrix");
  }
  {
    // Inner AMG hierarchies: A_Neu (LOBPCG preconditioner) first -- the eigensolve is waiting for it --
    // then A_Dir / A_Rob (local solves), whose host set-up runs on its own thread WHILE the GPU is busy
    // with the eigensolve; it is joined and uploaded when level 2 is done (or right away without level 2).
    lap("diagonals");
    auto ta = clk::now();
    // GenEO-2 runs the gamma eigenproblem through this hierarchy with whole blocks
    const int max_m1 = (opt.lvl2 == 2 && opt.els2_pc == "amg") ? eig_block_max() : 1;
    auto start1 = [this, ap1h, max_m1]() {
      Amg1Pending* pp = pend1.get();
      const std::vector<int> so = suboff;
      AmgParams ap = ap1h;
      ap.single = opt.dls1_amg_single;
      const bk::Csr* fine = &dirL;          // uploaded (with its companion) before this thread starts; outlives it
      pp->th = std::thread([pp, so, ap, max_m1, fine]() {
        auto t0 = clk::now();
        try {
          // Everything this thread launches goes to a private stream, concurrently with whatever the main stream is
          // running (the eigensolve).  Default: the sparse products of the hierarchy on the device (the host only
          // aggregates), as for the A_Neu hierarchy -- with ONE subdomain per GPU the host products of a 6.4 M-row matrix
          // took 1.07 s and were the critical path of the set-up.  GENEO_AMG1_HOST=1 (or a row beyond the product
          // kernels' capacity): host products + upload.
          bk::side_stream_begin();
          pp->dev = new AmgDevice();
          bool built = false;
          if (!getenv("GENEO_AMG1_HOST") && !getenv("GENEO_AMG_HOST")) {
            AmgLevelHostPart l0;
            const bool have0 = pp->level0.valid();
            if (have0) l0 = pp->level0.get();
            built = pp->dev->build_on_device(pp->mat(), so, ap, max_m1, fine, have0 ? &l0 : nullptr);
          }
          pp->res.secs = secs(t0, clk::now());
          pp->on_device = built;
          if (!built) {
            amg_setup_host(pp->mat(), so, ap, pp->res.levels, pp->res.cinv, pp->res.cbase);
            pp->res.secs = secs(t0, clk::now());
            auto t1 = clk::now();
            pp->dev->upload(pp->res.levels, pp->res.cinv, pp->res.cbase, ap, max_m1, fine);
            pp->upload_secs = secs(t1, clk::now());
          }
          bk::side_stream_end();
        } catch (std::exception& e) {
          bk::side_stream_end();
          pp->res.err = e.what();
        } catch (...) {
          bk::side_stream_end();
          pp->res.err = "GenEO: level-1 hierarchy set-up failed";
        }
        if (pp->res.secs == 0.0) pp->res.secs = secs(t0, clk::now());
      });
    };
    // The level-1 (A_Dir / A_Rob) hierarchy is not needed before the solve: host set-up on its own thread, joined
    // after the eigensolve.  The A_Neu hierarchy (LOBPCG waits for it) is built with the sparse products on the
    // device; the host only aggregates.  Fallback to the host products when a row exceeds the kernels' capacity.
    if (want1 && !wantN) start1();
    if (wantN) {
      try {
        amgN = new AmgDevice();
        bool built = false;
        if (!getenv("GENEO_AMG_HOST")) {
          AmgLevelHostPart l0;
          const bool have0 = pre_neu.valid();
          const auto t_w0 = clk::now();
          if (have0) l0 = pre_neu.get();
          if (getenv("GENEO_DEBUG")) fprintf(stderr, "[amg] waited %.3f s for the fine level's host part (diagonal, aggregates)\n", secs(t_w0, clk::now()));
          built = amgN->build_on_device(h_neuL, suboff, apN, eig_block_max(), &neuL, have0 ? &l0 : nullptr);
        }
        if (!built) {
          AmgHostResult rN;
          amg_setup_host(h_neuL, suboff, apN, rN.levels, rN.cinv, rN.cbase);
          amgN->upload(rN.levels, rN.cinv, rN.cbase, apN, eig_block_max(), &neuL);
        }
        info.amg_levels = amgN->nlevels();
        info.amg_operator_complexity = amgN->operator_complexity();
        info.amg_on_device = built ? 1 : 0;
      } catch (std::exception& e) {
        return fail(e.what());
      }
    }
    // the level-1 hierarchy starts once the A_Neu one (which the eigensolve is waiting for) is done: both are device
    // work now, and side by side they took 0.16 s instead of 0.11 s on the critical path (126^3)
    if (want1 && wantN) start1();
    bk::sync();
    info.amgSetupTime += secs(ta, clk::now());     // (+=: the groups of an overlapped set-up add theirs from another thread)
    if (getenv("GENEO_DEBUG")) fprintf(stderr, "[amg] A_Neu hierarchy (%s products) %.3f s\n", info.amg_on_device ? "device" : "host", info.amgSetupTime);
    if (want1 && !opt.lvl2)
      if (int rc = finish_amg1()) return rc;
  }
  bk::sync();
  info.lvl1SetupMinvTimeLoc = secs(t1, clk::now());
  is_setup = true;
  bk::set(d_x0, 0.0, n_owned());
  prepare_secs = secs(t0, clk::now());
  return 0;
}

// Second half: -geneo_chk diagnostics, level 2 (eigensolves, Z, E), bookkeeping.
int PC::setup_finish(const double* b_dev) {
  if (!is_setup) return fail("GenEO preconditioner: set-up not prepared");
  const auto t0 = clk::now();
  if (opt.check) {
    for (auto& s : subs)                      // geneo.cpp:988-997 (D = 1/mult, mult >= 1 validated on input)
      for (int v : s.mult)
        if (1.0 / (double)v <= DBL_EPSILON) return fail("GenEO - check D: bad partition of unity, min 0");
    try {
      if (int rc = check_global_spd()) { is_setup = false; return rc; }
    } catch (std::exception& e) {
      is_setup = false;
      return fail(e.what());
    }
  }
  if (opt.lvl2) {
    if (int rc = setup_level2(b_dev)) {
      is_setup = false;
      return rc;
    }
  }
  bk::sync();
  if (!eig_only) info.nullPivotsLoc = amg_null_pivots_take();
  if (info.nullPivotsLoc && getenv("GENEO_DEBUG"))
    fprintf(stderr, "[setup] %d null pivot(s) detected and fixed in the coarsest blocks of the local hierarchies (singular subdomain matrix)\n", info.nullPivotsLoc);
  info.setupTime = prepare_secs + secs(t0, clk::now());
  if (getenv("GENEO_DEBUG")) {
    double as = 0, fs = 0;
    long long na = 0;
    bk::alloc_stats(&as, &fs, &na);
    fprintf(stderr, "[setup] total %.3f s; since the previous report: %lld hipMalloc %.3f s, hipFree %.3f s\n", info.setupTime, na, as, fs);
  }
  return 0;
}

// ------------------------------------------------------------------------------------ R, R^T, comm
void PC::allreduce(double* dev, int n) {
  if (size == 1 || n == 0) return;
  for (int off = 0; off < n; off += comm_red_cap) {
    const int c = std::min(comm_red_cap, n - off);
    bk::d2d(comm_red, dev + off, sizeof(double) * c);
    if (cb_allreduce(cb_user, c)) throw std::runtime_error("GenEO: allreduce callback failed");
    bk::d2d(dev + off, comm_red, sizeof(double) * c);
  }
}

void PC::restrict_to_local(const double* x, double* xL) {
  if (size == 1) {
    bk::gather(xL, x, d_l2e, nL);
    return;
  }
  const int nown = n_owned();
  bk::gather(comm_send, x, d_send_idx, (int)send_idx.size());
  if (cb_exchange(cb_user, 0)) throw std::runtime_error("GenEO: halo exchange callback failed");
  bk::copy(d_xe, x, nown);
  bk::copy(d_xe + nown, comm_recv, nH);
  bk::gather(xL, d_xe, d_l2e, nL);
}

void PC::prolong_add(const double* wL, double* y) {
  const int nown = n_owned();
  if (size == 1) {
    bk::segsum(y, wL, d_rt_ptr, d_rt_idx, nown, false);
    return;
  }
  bk::segsum(d_ye, wL, d_rt_ptr, d_rt_idx, nE, false);
  bk::copy(comm_send, d_ye + nown, nH);
  if (cb_exchange(cb_user, 1)) throw std::runtime_error("GenEO: halo exchange callback failed");
  bk::copy(y, d_ye, nown);
  bk::segsum(y, comm_recv, d_rv_ptr, d_rv_idx, nown, true);
}

int PC::matmult(const double* x, double* y) {
  if (!is_setup) return fail("GenEO preconditioner is not set up");
  info.spmv_calls++;
  if (size == 1) {
    bk::spmv(neuE, x, d_wL);
  } else {
    const int nown = n_owned();
    bk::gather(comm_send, x, d_send_idx, (int)send_idx.size());
    if (cb_exchange(cb_user, 0)) return fail("GenEO: halo exchange callback failed");
    bk::copy(d_xe, x, nown);
    bk::copy(d_xe + nown, comm_recv, nH);
    bk::spmv(neuE, d_xe, d_wL);
  }
  prolong_add(d_wL, y);
  return 0;
}

// ---- the same three operators on row-major blocks of w vectors (blocked assembly of E) ----------------
// The halo buffers hold comm_width x the single-vector counts; the exchange callback gets the width in
// the upper bits of its flag (flag = reverse | width << 1).
void PC::restrict_block(const double* X, double* XL, int w, double* xe) {
  if (size == 1) {
    bk::gather_rows(XL, X, d_l2e, nL, w);
    return;
  }
  const int nown = n_owned();
  bk::gather_rows(comm_send, X, d_send_idx, (int)send_idx.size(), w);
  if (cb_exchange(cb_user, 0 | (w << 1))) throw std::runtime_error("GenEO: halo exchange callback failed");
  bk::copy(xe, X, (int)((int64_t)nown * w));
  bk::copy(xe + (int64_t)nown * w, comm_recv, (int)((int64_t)nH * w));
  bk::gather_rows(XL, xe, d_l2e, nL, w);
}

void PC::prolong_block(const double* WL, double* Y, int w, double* ye) {
  const int nown = n_owned();
  if (size == 1) {
    bk::segsum_rows(Y, WL, d_rt_ptr, d_rt_idx, nown, w, false);
    return;
  }
  bk::segsum_rows(ye, WL, d_rt_ptr, d_rt_idx, nE, w, false);
  bk::copy(comm_send, ye + (int64_t)nown * w, (int)((int64_t)nH * w));
  if (cb_exchange(cb_user, 1 | (w << 1))) throw std::runtime_error("GenEO: halo exchange callback failed");
  bk::copy(Y, ye, (int)((int64_t)nown * w));
  bk::segsum_rows(Y, comm_recv, d_rv_ptr, d_rv_idx, nown, w, true);
}

void PC::matmult_block(const double* X, double* Y, int w, double* WL, double* xe) {
  info.spmv_calls += w;
  if (size == 1) {
    bk::spmm_strided(neuE, X, w, WL, w, w, nullptr, nullptr);
  } else {
    const int nown = n_owned();
    bk::gather_rows(comm_send, X, d_send_idx, (int)send_idx.size(), w);
    if (cb_exchange(cb_user, 0 | (w << 1))) throw std::runtime_error("GenEO: halo exchange callback failed");
    bk::copy(xe, X, (int)((int64_t)nown * w));
    bk::copy(xe + (int64_t)nown * w, comm_recv, (int)((int64_t)nH * w));
    bk::spmm_strided(neuE, xe, w, WL, w, w, nullptr, nullptr);
  }
  prolong_block(WL, Y, w, xe);
}

// [D] M^-1 [D] on the concatenated local space: one independent Jacobi-PCG per subdomain,
// all subdomains advanced by the same launches (geneo.cpp:1991-2002 with MUMPS replaced).
void PC::local_solve(double* wL) {
  if (opt.lvl1RAS) bk::xmy(wL, wL, d_D, nL);
  const int ns = (int)subs.size();
  const bool use_amg = (opt.dls1_pc == "amg") && amg1;
  double* x = d_xL;  // solution
  const double* dinv = use_amg ? nullptr : d_dinv1;
  bk::cg_start(ch, d_cg_sc, x, d_cg_r, d_cg_z, d_cg_p, wL, dinv);
  if (use_amg) {
    amg1->vcycle(d_cg_r, 1, d_cg_z, 1, 1);
    bk::seg_partial(ch, d_cg_r, d_cg_z, 1);
    bk::cg_set_rz(ch, d_cg_sc);
    bk::copy(d_cg_p, d_cg_z, nL);
  }
  const double tol2 = opt.dls1_rtol * opt.dls1_rtol;
  std::vector<double> sc((size_t)8 * std::max(1, ns));
  int it = 0;
  int parity = 0;
  static const int amg_check = getenv("GENEO_DLS1_AMG_CHECK") ? atoi(getenv("GENEO_DLS1_AMG_CHECK")) : 4;
  const int check = std::max(1, use_amg ? std::min(amg_check, opt.dls1_check) : opt.dls1_check);
  bool done = false;
  // One chunk = `len` PCG iterations with device-resident scalars (~19 small launches each with the fused V-cycle):
  // launch-bound, so a chunk is captured once per length into a HIP graph and replayed.  An even chunk length brings
  // the rz parity back to 0, which makes every chunk of that length the same launch sequence.  Converged subdomains
  // freeze themselves on the device (k_cg_flag), so a chunk may run past a subdomain's convergence.
  auto chunk = [&](int len) {
    for (int k = 0; k < len; ++k) {
      bk::spmv(dirL, d_cg_p, d_cg_q);
      bk::seg_pap(ch, d_cg_p, d_cg_q);
      bk::cg_update(ch, d_cg_sc, parity, x, d_cg_r, d_cg_z, d_cg_p, d_cg_q, dinv);
      if (use_amg) {
        amg1->vcycle(d_cg_r, 1, d_cg_z, 1, 1);
        bk::seg_partial(ch, d_cg_r, d_cg_z, 1);
      }
      bk::cg_direction(ch, d_cg_sc, parity, d_cg_p, d_cg_z, tol2);
      parity ^= 1;
    }
  };
  // (a chunk's launches carry the rz parity it starts with: graphs are kept per (length, parity); odd lengths other than
  // the single step of the tail below are not captured)
  auto graph_for = [&](int len) -> void* {
    if ((len % 2 != 0 && len != 1) || cg_graph_failed) return nullptr;
    const int key = 2 * len + parity;
    auto it_g = cg_graphs.find(key);
    if (it_g != cg_graphs.end()) return it_g->second;
    void* g = nullptr;
    const int parity_in = parity;
    if (bk::graph_capture_begin()) {
      try {
        chunk(len);
      } catch (...) {       // leave the capture cleanly (the stream and the capture flag are global state)
        bk::graph_capture_end();
        cg_graph_failed = true;
        throw;
      }
      g = bk::graph_capture_end();
      parity = parity_in;        // recorded, not run
    }
    if (!g) cg_graph_failed = true;
    else cg_graphs[key] = g;
    return g;
  };
  static const bool dbg_dls1 = getenv("GENEO_DEBUG_DLS1") != nullptr;   // per solve: the chunk boundary each subdomain froze at
  std::vector<int> conv_at(dbg_dls1 ? ns : 0, -1);
  // runs `len` iterations, then reads the per-subdomain flags back
  auto run = [&](int len) {
    void* g = graph_for(len);
    // While bench.py's in-situ kernel timer runs, every 8th chunk goes out as direct launches: HIP events cannot bracket
    // kernels inside a replayed graph (hipEventElapsedTime rejects events recorded by graph nodes), the other chunks
    // replay the graph as they do outside the benchmark.
    const bool direct = !g || (bk::spmv_profiling() && (cg_chunks++ % 8 == 0));
    if (!direct) {
      bk::graph_launch(g);
      parity ^= (len & 1);
    } else {
      chunk(len);
    }
    it += len;
    bk::d2h(sc.data(), d_cg_sc, sizeof(double) * 8 * ns);
    done = true;
    for (int s = 0; s < ns; ++s) {
      if (sc[(size_t)s * 8 + 6] != 0.0) done = false;
      else if (dbg_dls1 && conv_at[s] < 0) conv_at[s] = it;
    }
  };
  // Every chunk boundary is a host round trip (flags D2H + graph launch: ~0.17 ms against 0.3 ms per iteration).  The
  // solves of one set-up need almost the same number of iterations (126^3: 17 to 20), so after the first solve the
  // others start with ONE long chunk just below that number and finish in chunks of 2.
  static const bool adaptive = !getenv("GENEO_DLS1_FIXED_CHUNKS");
  if (use_amg && adaptive && cg_long_len >= 8 && check % 2 == 0) {
    const int first = cg_long_len;
    run(first);
    // behind the long chunk: pairs of iterations, or -- where an iteration costs far more than the host round trip of a
    // chunk boundary (4 M local rows and more: 0.5 ms and up against 0.17 ms) -- single ones, so that a solve stops at the
    // iteration that converged it instead of the next even one
    const int64_t single_rows = getenv("GENEO_DLS1_SINGLE_STEP_ROWS") ? atoll(getenv("GENEO_DLS1_SINGLE_STEP_ROWS")) : 4000000;   // (per solve: tests)
    const int tail = (int64_t)nL >= single_rows ? 1 : 2;
    while (!done && it < opt.dls1_max_it) run(tail);
    // The long chunk follows what the solves need.  The first solve of a set-up (the one the length was taken from) is the
    // hardest -- 368^3: 24 iterations, the others 16 -- and a chunk that ends after convergence is iterations nobody asked
    // for: at 5.6 ms each on 52 M rows, two of them per solve are 10 % of the solve.  Converged inside the long chunk: probe
    // a chunk two shorter next time; two or more short chunks behind it: two longer.  Steady state: the long chunk plus one
    // short one, i.e. the iterations needed rounded up to even.
    if (it == first && first > 8) cg_long_len = first - 2;
    else if (it >= first + 4) cg_long_len = first + 2;
  } else {
    while (!done && it < opt.dls1_max_it) run(check);
    if (use_amg && adaptive && done && cg_long_len == 0 && check % 2 == 0) {
      const int len = ((it - check - 2) / 2) * 2;       // the solve needed more than it - check iterations
      cg_long_len = len >= 8 ? len : -1;                 // -1: too short to bother
    }
  }
  info.dls1_iterations += it;
  info.dls1_solves += 1;
  if (dbg_dls1) {
    fprintf(stderr, "[dls1] solve %lld: %d iterations (first chunk %d), subdomains frozen at", info.dls1_solves, it, cg_long_len);
    for (int s = 0; s < ns; ++s) fprintf(stderr, " %d", conv_at[s]);
    fprintf(stderr, "\n");
  }
  if (!done) throw std::runtime_error("GenEO - solve KO: dls1 (KSP_DIVERGED_ITS)");
  if (opt.lvl1SRAS) bk::xmy(wL, x, d_D, nL);
  else bk::copy(wL, x, nL);
}

// yE = E^-1 Z^T x, Z^T x taken from the already restricted xL; replicated host solve
void PC::coarse_solve_local(const double* xL, double* yE) {
  auto t0 = clk::now();
  bk::zt_apply(ch, d_Z, d_zbase, d_ksub, d_zoff, kmax, xL, yE, dimE);
  allreduce(yE, dimE);
  auto t1 = clk::now();
  // E^-1 (geneo.cpp:1493, KSPSolve(pcKSPL2)): the replicated Cholesky factor lives on the device and the two triangular
  // sweeps are one launch behind the all-reduce -- no download, host solve, upload and no host synchronisation in the
  // preconditioner application.  Host path: E not positive definite to rounding (LU with pivoting) or dimE > 1024.
  if (!(E_chol && d_EL && bk::chol_solve(d_EL, d_ELT, dimE, yE))) {
    bk::d2h(h_yE.data(), yE, sizeof(double) * dimE);
    if (E_chol) dense::cholesky_solve_lu(Efac, EfacT, dimE, h_yE.data());
    else dense::lu_solve(Efac, dimE, Epiv, h_yE.data());
    bk::h2d(yE, h_yE.data(), sizeof(double) * dimE);
  }
  auto t2 = clk::now();
  info.lvl2ApplyZtTimeLoc += secs(t0, t1);
  info.lvl2ApplyEinvTimeLoc += secs(t1, t2);
}

int PC::apply_q(const double* x, double* y) {
  if (!is_setup || !opt.lvl2) return fail("GenEO preconditioner: no coarse space");
  try {
    restrict_to_local(x, d_xL);
    coarse_solve_local(d_xL, d_yE);
    bk::z_apply(ch, d_Z, d_zbase, d_ksub, d_zoff, d_yE, d_wL, false);
    prolong_add(d_wL, y);
  } catch (std::exception& e) {
    return fail(e.what());
  }
  return 0;
}

int PC::apply(const double* x, double* y) {
  if (!is_setup) return fail("GenEO preconditioner is not set up");
  const int nown = n_owned();
  try {
    auto t0 = clk::now();
    if (opt.lvl2 && !opt.hybrid) {
      // y = sum R^T ( Z_s E^-1 Z^T x + [D] M^-1 [D] R x ): one restriction, one prolongation
      restrict_to_local(x, d_wL);
      coarse_solve_local(d_wL, d_yE);
      auto t1 = clk::now();
      local_solve(d_wL);
      auto t2 = clk::now();
      bk::z_apply(ch, d_Z, d_zbase, d_ksub, d_zoff, d_yE, d_wL, true);
      prolong_add(d_wL, y);
      info.lvl2ApplyTimeLoc += secs(t0, t1);
      info.lvl1ApplyMinvTimeLoc += secs(t1, t2);
      info.lvl1ApplyTimeLoc += secs(t1, clk::now());
      return 0;
    }
    // generic composition (geneo.cpp:2074-2094)
    bool have_q = false;
    if (opt.lvl2 && !opt.effHybrid) {  // applyLevel2
      if (int rc = apply_q(x, y)) return rc;
      have_q = true;
      info.lvl2ApplyTimeLoc += secs(t0, clk::now());
    }
    auto t1 = clk::now();
    double* w = d_t1;
    bk::copy(w, x, nown);
    if (opt.hybrid && !opt.effHybrid) {  // (I - P^T): w = x - A (Q x), geneo.cpp:1931,1944
      if (int rc = matmult(y, d_t2)) return rc;
      bk::axpy(w, -1.0, d_t2, nown);
    }
    restrict_to_local(w, d_wL);
    auto t2 = clk::now();
    local_solve(d_wL);
    info.lvl1ApplyMinvTimeLoc += secs(t2, clk::now());
    prolong_add(d_wL, w);
    if (opt.hybrid) {  // (I - P): w = w - Q (A w), geneo.cpp:1935-1944
      if (int rc = matmult(w, d_t2)) return rc;
      if (int rc = apply_q(d_t2, d_t3)) return rc;
      bk::axpy(w, -1.0, d_t3, nown);
    }
    if (have_q) bk::axpy(y, 1.0, w, nown);
    else bk::copy(y, w, nown);
    info.lvl1ApplyTimeLoc += secs(t1, clk::now());
  } catch (std::exception& e) {
    return fail(e.what());
  }
  return 0;
}

// ------------------------------------------------------------------------------------ level 2
// the local eigensolves of level 2 (buildCoarseSpaceWithGenEO, geneo.cpp:1243-1366): eigenvalues, kept counts, Z
int PC::setup_level2_eigen() {
  const int ns = (int)subs.size();
  auto t0 = clk::now();
  eigvals.assign(ns, {});
  candidates.assign(ns, {});
  local_tau();
  if (opt.lvl2 != 2) tauLoc.assign(ns, opt.tau);  // GenEO-1 uses the global tau (geneo.cpp:1271)
  int nmax = 0;
  for (auto& s : subs) nmax = std::max(nmax, (int)s.l2g.size());
  int rc = 0;
  try {
    if (nmax <= 192) rc = eigen_dense_host();
    else if (eig_groups.size() > 2) rc = eigen_grouped();
    else rc = eigen_lobpcg();
  } catch (std::exception& e) {
    return fail(e.what());
  }
  if (rc) return rc;
  bk::sync();
  info.lvl2SetupEigTimeLoc = secs(t0, clk::now());
  return 0;
}

int PC::setup_level2(const double* b_dev) {
  if (!eig_early)             // (PC::setup has run them already, next to this PC's own device preparation)
    if (int rc = setup_level2_eigen()) return rc;
  auto t1 = clk::now();
  if (eig_only) return 0;      // a group of eigen_grouped: Z, the eigenvalues and the counters are what the owner takes
  if (int r1 = finish_amg1()) return r1;
  t1 = clk::now();
  try {
    if (opt.check)
      if (int rc = check_local_rank()) return rc;
    if (int r2 = build_E()) return r2;
    if (opt.check)
      if (int rc = check_global_rank()) return rc;
  } catch (std::exception& e) {
    return fail(e.what());
  }
  bk::sync();
  info.lvl2SetupETimeLoc = secs(t1, clk::now());
  // initial guess (geneo.cpp:1601-1607)
  if (opt.effHybrid && b_dev) {
    if (int r3 = apply_q(b_dev, d_x0)) return r3;
  }
  return 0;
}

// Small subdomains (n <= 192, e.g. the reference's tst/dummy cases): the projected problem IS the
// full pencil; solved with the same rank-revealing Rayleigh-Ritz routine LOBPCG uses.
int PC::eigen_dense_host() {
  const int ns = (int)subs.size();
  const bool g2 = (opt.lvl2 == 2);
  int nev_try = 0;
  eig_targets(&nev_try);
  if (g2)
    if (int rc = local_gamma()) return rc;
  std::vector<std::vector<std::vector<double>>> vecs(ns);
  auto densify = [](const HostCsr& a, const std::vector<int>* mult, std::vector<double>& G) {
    const int n = a.n;
    G.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
      for (int k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
        const int c = a.col[k];
        G[(size_t)i * n + c] += mult ? a.val[k] / ((double)(*mult)[i] * (double)(*mult)[c]) : a.val[k];
      }
  };
  for (int s = 0; s < ns; ++s) {
    Sub& sd = subs[s];
    const int n = (int)sd.l2g.size();
    std::vector<double> GA, GW, GR;
    densify(sd.a_neu, nullptr, GA);
    densify(sd.a_dir, &sd.mult, GW);           // B_w = D A_Dir D
    if (g2) {
      HostCsr rob;
      make_robin(sd, rob);
      densify(rob, nullptr, GR);
    }
    const std::vector<double>& GB = g2 ? GR : GW;
    if (opt.check) {  // checkSPD of the pencil's right-hand matrix (geneo.cpp:883-886), dense here
      for (const char* pb : {"tau", "gamma"}) {
        if (!g2 && pb[0] == 'g') break;
        std::vector<double> G = GB, w, V;
        dense::sym_eig(G, n, w, V);
        int neg = 0, nul = 0, pos = 0;
        double wmax = 0.0, wmin = 1e300;
        for (double v : w) { wmax = std::max(wmax, std::fabs(v)); wmin = std::min(wmin, v); }
        for (double v : w) {
          if (std::fabs(v) <= 1e-14 * wmax) nul++;
          else if (v < 0) neg++;
          else pos++;
        }
        std::ofstream f((check_id(sd.gid, nsub_global) + ".SPD." + pb + ".B.log").c_str());
        f << pb << ".B - eigen value 0: " << wmin << std::endl;
        f << std::endl << pb << ".B - inertia: nbNegEV " << neg << ", nbNullEV " << nul << ", nbPosEV " << pos << std::endl;
        if (std::fabs(wmin) <= DBL_EPSILON) {
          std::ostringstream msg;
          msg << "GenEO - check SPD: " << pb << ".B not SPD, bad eigen value " << wmin;
          return fail(msg.str());
        }
        if (neg > 0 || nul > 0) return fail("GenEO - check SPD: not SPD (inertia - negative or null eigen value found)");
      }
    }
    std::vector<double> th, C;
    int r = dense::gen_eig_rr(GA, GB, n, 0, 1e-13, th, C);
    // eigenLocalProblem :855-879 with the inertia count read off the full spectrum
    auto nev_for = [&](int est) {
      int nev = nev_try;
      if (!opt.noSyl && est > 0) {
        nev = est;
        const int cut = (g2 && opt.cut >= 2) ? opt.cut / 2 : opt.cut;
        if (cut > 0) nev = std::min(nev, cut);
      }
      return nev;
    };
    int est = 0;
    for (int j = 0; j < r; ++j) est += (th[j] < tauLoc[s]);
    int nev = std::min(std::min(nev_for(est), n), r);
    // TARGET_MAGNITUDE around 0 (geneo.cpp:638-640): smallest |theta| first
    std::vector<int> ord(r);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return std::fabs(th[a]) < std::fabs(th[b]); });
    for (int j = 0; j < nev; ++j) {
      const int c = ord[j];
      candidates[s].push_back(th[c]);
      if (th[c] > tauLoc[s]) continue;  // geneo.cpp:713
      eigvals[s].push_back(th[c]);
      std::vector<double> v(n);
      for (int i = 0; i < n; ++i) v[i] = C[(size_t)i * r + c];
      vecs[s].push_back(std::move(v));
    }
    // Nicolaides (geneo.cpp:897-944)
    if (!eigvals[s].empty() && *std::min_element(eigvals[s].begin(), eigvals[s].end()) >= opt.nicolaides_zero * DBL_EPSILON) {
      double num = 0.0, den = 0.0;
      for (size_t e = 0; e < GA.size(); ++e) { num += GA[e]; den += GB[e]; }
      if (std::fabs(num / den) <= FLT_EPSILON) {
        eigvals[s].push_back(0.0);
        vecs[s].push_back(std::vector<double>(n, 1.0));
        info.nicolaidesLoc++;
      }
    }
    if (g2) {  // gamma problem (geneo.cpp:1299): largest eigenvalues of B_w v = lambda A_Rob v, kept when >= gamma_loc
      r = dense::gen_eig_rr(GW, GR, n, 0, 1e-13, th, C);
      est = 0;
      for (int j = 0; j < r; ++j) est += (th[j] > gammaLoc[s]);
      nev = std::min(std::min(nev_for(est), n), r);
      ord.resize(r);
      std::iota(ord.begin(), ord.end(), 0);
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return std::fabs(th[a]) > std::fabs(th[b]); });
      for (int j = 0; j < nev; ++j) {
        const int c = ord[j];
        candidates[s].push_back(th[c]);
        if (th[c] < gammaLoc[s]) continue;  // geneo.cpp:717
        eigvals[s].push_back(th[c]);
        std::vector<double> v(n);
        for (int i = 0; i < n; ++i) v[i] = C[(size_t)i * r + c];
        vecs[s].push_back(std::move(v));
      }
    }
    if (vecs[s].empty()) {  // geneo.cpp:1305-1314
      eigvals[s].push_back(0.0);
      vecs[s].push_back(std::vector<double>(n, 1.0));
      info.nicolaidesLoc++;
    }
    info.estimDimELoc += (int)eigvals[s].size();
  }
  // Z_s = D .* v (fillZE2L, geneo.cpp:249-286), column-major per subdomain
  ksub.assign(ns, 0);
  std::vector<int64_t> zbase(ns + 1, 0);
  for (int s = 0; s < ns; ++s) {
    ksub[s] = (int)vecs[s].size();
    zbase[s + 1] = zbase[s] + (int64_t)ksub[s] * (int64_t)subs[s].l2g.size();
  }
  std::vector<double> Z((size_t)std::max<int64_t>(1, zbase[ns]));
  for (int s = 0; s < ns; ++s) {
    const int n = (int)subs[s].l2g.size();
    for (int j = 0; j < ksub[s]; ++j)
      for (int i = 0; i < n; ++i) Z[zbase[s] + (int64_t)j * n + i] = vecs[s][j][i] / (double)subs[s].mult[i];
  }
  d_Z = (double*)bk::alloc(sizeof(double) * Z.size());
  bk::h2d(d_Z, Z.data(), sizeof(double) * Z.size());
  d_zbase = (int64_t*)bk::alloc(sizeof(int64_t) * (ns + 1));
  bk::h2d(d_zbase, zbase.data(), sizeof(int64_t) * (ns + 1));
  return 0;
}

// Persistent host workers for the per-subdomain dense work inside the LOBPCG loop (Rayleigh-Ritz, Gram propagation):
// creating and joining eight std::threads costs ~0.18 ms per round on the GPU box's host, two rounds per iteration.
// run(s0, s1, f) calls f(s) for every s in [s0, s1) on the workers plus the calling thread and returns when all are done.
class HostPool {
 public:
  static HostPool& get() {
    static HostPool p;
    return p;
  }
  void run(int s0, int s1, const std::function<void(int)>& f) {
    if (s1 - s0 <= 1 || workers.empty()) {
      for (int s = s0; s < s1; ++s) f(s);
      return;
    }
    // a nested run() from inside a work item runs its items inline: the pool is busy with the outer list, and the thread
    // that owns run_mu (the outer caller, working through its share) must not try to lock it a second time
    if (in_pool_work) {
      for (int s = s0; s < s1; ++s) f(s);
      return;
    }
    // one work list at a time: a second PC setting up on another host thread waits here
    std::unique_lock<std::mutex> one(run_mu);
    std::exception_ptr err;
    {
      std::unique_lock<std::mutex> lk(mu);
      fn = &f;
      next = s0;
      end = s1;
      pending = s1 - s0;
      perr = &err;
      ++generation;
    }
    cv.notify_all();
    work();                                   // the caller takes its share
    std::unique_lock<std::mutex> lk(mu);
    done_cv.wait(lk, [&] { return pending == 0; });
    fn = nullptr;
    if (err) std::rethrow_exception(err);
  }
  ~HostPool() {
    {
      std::unique_lock<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : workers) t.join();
  }

 private:
  HostPool() {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n = (int)std::min(15u, hw > 1 ? hw - 1 : 0u);
    for (int i = 0; i < n; ++i)
      workers.emplace_back([this]() {
        unsigned long long seen = 0;
        for (;;) {
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
          }
          work();
        }
      });
  }
  void work() {
    for (;;) {
      int s;
      const std::function<void(int)>* f;
      {
        std::unique_lock<std::mutex> lk(mu);
        if (!fn || next >= end) return;
        s = next++;
        f = fn;
      }
      in_pool_work = true;
      try {
        (*f)(s);
      } catch (...) {
        std::unique_lock<std::mutex> lk(mu);
        if (perr && !*perr) *perr = std::current_exception();
      }
      in_pool_work = false;
      std::unique_lock<std::mutex> lk(mu);
      if (--pending == 0) done_cv.notify_all();
    }
  }
  std::vector<std::thread> workers;
  std::mutex run_mu;
  static thread_local bool in_pool_work;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  const std::function<void(int)>* fn = nullptr;
  std::exception_ptr* perr = nullptr;
  int next = 0, end = 0, pending = 0;
  unsigned long long generation = 0;
  bool stop = false;
};

thread_local bool HostPool::in_pool_work = false;

// LOBPCG on a pencil A v = lambda B v (lowest eigenvalues) for all local subdomains in lock step.
// Basis S = [X | P | W] (n_L x 3m row-major).  Per iteration and subdomain:
//   W = T (A X - B X Lambda)              T = AMG V-cycle of A, or Chebyshev(degree, Jacobi) on A   (SpMM)
//   G_A = S^T (A S), G_B = S^T (B S)      FP64-MFMA Gram kernels
//   rank-revealing Rayleigh-Ritz on host  (3m x 3m)
//   [X P] <- S C, same for A S, B S       FP64-MFMA block update kernels
// On return lam[s*m+j] holds the Ritz values (1e300 = fewer independent directions than m) and Xc
// (n_L x m, row-major, device) the B-orthonormal Ritz vectors.
int PC::lobpcg_solve(const EigProblem& P, int m, std::vector<double>& lam, double* Xc) {
  const int ns = (int)subs.size();
  // the rows the pencil lives on: the rank's fine rows, or those of a multigrid level (coarse start)
  const int nL = P.rows >= 0 ? P.rows : this->nL;
  const bk::Chunks& ch = P.chunks ? *P.chunks : this->ch;
  auto rows_of = [&](int s) { return P.row_off ? P.row_off[s + 1] - P.row_off[s] : (int)subs[s].l2g.size(); };
  const int nev_try = P.nev_try;
  const int p3 = 3 * m;
  const size_t blk = (size_t)nL * p3;
  auto dv = [](size_t n) { return (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, n)); };
  double *S = dv(blk), *AS = dv(blk), *BS = dv(blk), *T = dv(blk), *AT = dv(blk), *BT = dv(blk);
  double *cr = dv((size_t)nL * m), *cd = dv((size_t)nL * m), *cad = dv((size_t)nL * m);
  double *dGA = dv((size_t)ns * p3 * p3), *dGB = dv((size_t)ns * p3 * p3), *dC = dv((size_t)ns * p3 * 2 * m);
  double* dGW = dv((size_t)ns * 2 * m * p3);
  double *dlam = dv((size_t)ns * m), *dnr = dv((size_t)ns * m), *dna = dv((size_t)ns * m), *dnb = dv((size_t)ns * m);
  std::vector<double*> owned_bufs = {S, AS, BS, T, AT, BT, cr, cd, cad, dGA, dGB, dGW, dC, dlam, dnr, dna, dnb};
  void** graphs_to_free = nullptr;
  bk::Chunks* groups_to_free = nullptr;
  void** events_to_free = nullptr;
  auto cleanup = [&]() {
    for (double* p : owned_bufs) bk::dfree(p);
    for (int i = 0; i < 2; ++i) {
      if (groups_to_free && groups_to_free[i].start) bk::chunks_free(groups_to_free[i]);
      if (events_to_free && events_to_free[i]) bk::event_destroy(events_to_free[i]);
    }
    if (graphs_to_free)
      for (int i = 0; i < 2; ++i) bk::graph_destroy(graphs_to_free[i]);
  };

  auto applyA = [&](const double* X, double* Y) { bk::spmm_strided(*P.A, X, p3, Y, p3, m, P.As, P.As); info.eig_spmm++; };
  auto applyB = [&](const double* X, double* Y) { bk::spmm_strided(*P.B, X, p3, Y, p3, m, P.Bs, P.Bs); info.eig_spmm++; };

  std::vector<double> hGA((size_t)ns * p3 * p3), hGB((size_t)ns * p3 * p3);
  // what moves between host and device in EVERY iteration lives in page-locked memory (one DMA per copy, uploads without a
  // host synchronisation): the Rayleigh-Ritz coefficients, keep / lam of the fused update, the W rows of the Gram matrices
  struct Pinned {
    double* p = nullptr;
    size_t n = 0;
    explicit Pinned(size_t count) : p((double*)bk::pinned_alloc(sizeof(double) * std::max<size_t>(1, count))), n(count) {
      std::fill(p, p + n, 0.0);
    }
    ~Pinned() { bk::pinned_free(p); }
    Pinned(const Pinned&) = delete;
    Pinned& operator=(const Pinned&) = delete;
    double* data() { return p; }
    double* begin() { return p; }
  };
  Pinned hC((size_t)ns * p3 * 2 * m), hKL((size_t)ns * 2 * m);
  Pinned hGw((size_t)ns * 2 * m * p3);   // W rows of both Gram matrices: [(A W)^T S ; (B W)^T S] per subdomain
  bool have_prop = false;
  bool fused_update = false;  // set below, once the convergence test is known (m = 32, shift-invert test, MFMA on)
  bool have_R = false;        // the residual block of the current X is already in `cr` (written by the fused update)
  // "Lean" iteration (round 3, default where A and B share a sliced pattern): A S and B S are NOT carried.  With the Gram
  // blocks of [X P] propagated on the host, the only consumers of A X, B X, A P, B P were the recurrences that produce
  // them and the residual A X - B X diag(lam); the residual now comes from one two-operator product over X'
  // (bk::spmm_dual_residual: both products in registers, only R written) and the update moves S alone: 1.28 + 0.6 KB per
  // row and iteration instead of 4.0.  The residual is the true one in every iteration (no recurrence drift); the explicit
  // 96 x 96 Gram refresh computes A [X P], B [X P] when it needs them.
  bool lean = false;
  std::vector<double> keep((size_t)ns * m, 1.0);
  double* dkeep = nullptr;    // allocated below with the other per-pair device arrays
  static const bool full_gram = getenv("GENEO_LOBPCG_FULL_GRAM") != nullptr;   // experiment: explicit 96 x 96 Grams always
  std::vector<double> nr((size_t)ns * m), na((size_t)ns * m), nb((size_t)ns * m);
  std::vector<std::vector<double>> res(ns, std::vector<double>(m, 1.0));
  lam.assign((size_t)ns * m, 0.0);

  // Deflated restart: Q <- Q - Y_b (BY_b^T Q) for every locked block (Q: n_L x m inside a block of leading dimension ld).
  // X and P are combinations of earlier [X P W], so projecting the start block and every W keeps the whole basis in the
  // B-orthogonal complement of the locked vectors.
  double* dGd = nullptr;
  if (P.defl && !P.defl->empty()) {
    int kmax_d = 0;
    for (auto& b : *P.defl) kmax_d = std::max(kmax_d, b.k);
    dGd = dv((size_t)ns * kmax_d * m);
    owned_bufs.push_back(dGd);
  }
  auto deflate = [&](double* Q, int ld) {
    if (!dGd) return;
    for (auto& b : *P.defl) {
      bk::gram(ch, b.BY, b.k, b.k, Q, ld, m, dGd);
      bk::axpby(dGd, -1.0, dGd, 0.0, ns * b.k * m);
      bk::block_mul(ch, b.Y, b.k, b.k, dGd, m, Q, ld, true);
    }
  };
  // ---- start block + Rayleigh-Ritz on X alone
  if (P.X0) bk::block_axpby(S, p3, 1.0, P.X0, m, 0.0, nL, m);
  else bk::block_init(ch, S, p3, m, d_subgid, opt.eps_seed + (uint64_t)P.seed_off);
  deflate(S, p3);
  applyA(S, AS);
  applyB(S, BS);
  std::vector<char> frozen(ns, 0), locked((size_t)ns * m, 0);
  if (P.skip)
    for (int s = 0; s < ns; ++s) frozen[s] = P.skip[s] ? 1 : 0;
  std::vector<double> mask((size_t)ns * m, 1.0);
  double* dmask = dv((size_t)ns * m);
  owned_bufs.push_back(dmask);
  bk::h2d(dmask, mask.data(), sizeof(double) * mask.size());   // all ones: the device phase always takes the mask
  dkeep = dv((size_t)ns * m);
  owned_bufs.push_back(dkeep);
  void* it_graph[2] = {nullptr, nullptr};                       // HIP graphs of the device phase, one per buffer parity
  bool graph_has_gram[2] = {true, true};
  // two groups of subdomains for the pipelined host part (see `pipe` in the loop): chunk lists and Gram events
  // (opt-in, GENEO_LOBPCG_PIPELINE=1: measured on the 126^3 bench it shortens the GPU's wait for the host by 60 ms per
  // eigensolve and gives as much back in host-side synchronisation -- see DESIGN.md section 7)
  static const bool want_pipeline = getenv("GENEO_LOBPCG_PIPELINE") != nullptr;
  const bool pipeline_ok = ns >= 2 && want_pipeline;
  int grp[3] = {0, ns, ns};
  bk::Chunks chg[2];
  void* gram_ev[2] = {nullptr, nullptr};
  if (pipeline_ok) {
    std::vector<int> so(ns + 1, 0);
    for (int s = 0; s < ns; ++s) so[s + 1] = so[s] + rows_of(s);
    int cut = 1;                                                 // the split that balances the rows of the two groups
    for (int s = 1; s < ns; ++s)
      if (std::abs(2 * (int64_t)so[s] - so[ns]) < std::abs(2 * (int64_t)so[cut] - so[ns])) cut = s;
    grp[1] = cut;
    chg[0] = bk::chunks_upload(cut, so.data());
    chg[1] = bk::chunks_upload(ns - cut, so.data() + cut);
    gram_ev[0] = bk::event_create();
    gram_ev[1] = bk::event_create();
  }
  groups_to_free = chg;
  events_to_free = gram_ev;
  bool it_graph_failed = false;
  static const bool no_graph = getenv("GENEO_LOBPCG_NO_GRAPH") != nullptr;
  graphs_to_free = it_graph;
  double* dn3 = dv((size_t)ns * 3 * m);
  owned_bufs.push_back(dn3);
  std::vector<double> n3((size_t)ns * 3 * m);
  double t_rr_host = 0.0, t_dev_wait = 0.0, t_prop_host = 0.0;
  bool conv_sinvert_now = false;   // set with conv_sinvert below: the propagated Gram blocks are only used on that path
  auto t_lob0 = clk::now();
  // Gram blocks of the leading p columns, symmetrised on the host
  auto gram_blocks = [&](int p) {
    bk::gram(ch, S, p3, p, AS, p3, p, dGA);
    bk::gram(ch, S, p3, p, BS, p3, p, dGB);
    auto tg0 = clk::now();
    bk::d2h(hGA.data(), dGA, sizeof(double) * (size_t)ns * p * p);
    bk::d2h(hGB.data(), dGB, sizeof(double) * (size_t)ns * p * p);
    t_dev_wait += secs(tg0, clk::now());
  };
  // The Gram blocks of the NEW [X P] follow from the old ones: [X' P'] = S C  =>  [X' P']^T A [X' P'] = C^T G_A C.
  // Only the W rows of the next Gram matrices then need the GPU (a third of the flops, 5 block passes instead of
  // 12); the explicit 96 x 96 products come back every `refresh` iterations together with the explicit A X, B X.
  // The products run on the host AFTER the device part of the next iteration has been launched (run_propagate below):
  // they are needed only when its Gram rows come back, so they cost no GPU idle time.
  std::vector<double> hSA((size_t)ns * p3 * p3), hSB((size_t)ns * p3 * p3);   // symmetrised blocks the last RR used
  bool prop_pending = false;
  auto on_host_threads = [&](const std::function<void(int)>& f, int s0 = 0, int s1 = -1) {
    HostPool::get().run(s0, s1 < 0 ? ns : s1, f);
  };
  auto run_propagate = [&]() {
    if (!prop_pending) return;
    prop_pending = false;
    auto tg = clk::now();
    const int p = p3, qout = 2 * m;
    on_host_threads([&](int sd) {
      const double* c = hC.data() + (size_t)sd * p * qout;
      std::vector<double> tmp((size_t)p * qout);
      for (int which = 0; which < 2; ++which) {
        const double* g = (which ? hSB : hSA).data() + (size_t)sd * p * p;
        double* out = (which ? hGB : hGA).data() + (size_t)sd * p3 * p3;
        std::fill(out, out + (size_t)p3 * p3, 0.0);
        for (int i = 0; i < p; ++i) {            // tmp = G C
          double* ti = tmp.data() + (size_t)i * qout;
          std::fill(ti, ti + qout, 0.0);
          for (int k = 0; k < p; ++k) {
            const double gik = g[(size_t)i * p + k];
            if (gik == 0.0) continue;
            const double* ck = c + (size_t)k * qout;
            for (int j = 0; j < qout; ++j) ti[j] += gik * ck[j];
          }
        }
        for (int k = 0; k < p; ++k) {            // out[0:qout, 0:qout] = C^T tmp
          const double* ck = c + (size_t)k * qout;
          const double* tk = tmp.data() + (size_t)k * qout;
          for (int a = 0; a < qout; ++a) {
            const double cka = ck[a];
            if (cka == 0.0) continue;
            double* oa = out + (size_t)a * p3;
            for (int b = 0; b < qout; ++b) oa[b] += cka * tk[b];
          }
        }
      }
    });
    t_prop_host += secs(tg, clk::now());
  };
  // Subdomains [s0, s1) (all of them by default) with the chunk list `cg` of exactly those rows; `last`: the basis
  // buffers swap roles once the last group has been updated.
  auto rayleigh_ritz = [&](int p, int nfix, int qout, bool with_p, int s0 = 0, int s1 = -1, const bk::Chunks* cg = nullptr,
                           bool last = true) -> int {
    if (s1 < 0) s1 = ns;
    if (!cg) cg = &ch;
    auto tg1 = clk::now();
    std::fill(hC.begin() + (size_t)s0 * p * qout, hC.begin() + (size_t)s1 * p * qout, 0.0);
    const bool want_prop = (p == p3) && conv_sinvert_now;
    auto rr_one = [&](int s) {
      std::vector<double> ga(hGA.begin() + (size_t)s * p * p, hGA.begin() + (size_t)(s + 1) * p * p);
      std::vector<double> gb(hGB.begin() + (size_t)s * p * p, hGB.begin() + (size_t)(s + 1) * p * p);
      for (int a = 0; a < p; ++a)
        for (int b = a + 1; b < p; ++b) {
          ga[a * p + b] = ga[b * p + a] = 0.5 * (ga[a * p + b] + ga[b * p + a]);
          gb[a * p + b] = gb[b * p + a] = 0.5 * (gb[a * p + b] + gb[b * p + a]);
        }
      if (want_prop) {
        std::copy(ga.begin(), ga.end(), hSA.begin() + (size_t)s * p * p);
        std::copy(gb.begin(), gb.end(), hSB.begin() + (size_t)s * p * p);
      }
      double* cs = hC.data() + (size_t)s * p * qout;
      if (frozen[s]) {  // converged subdomain: keep X, drop P (identity update)
        for (int j = 0; j < m; ++j) cs[(size_t)j * qout + j] = 1.0;
        return;
      }
      std::vector<double> th, C;
      const int r = dense::gen_eig_rr(ga, gb, p, nfix, opt.rr_drop, th, C);
      for (int j = 0; j < m; ++j) {
        if (j < r) {
          lam[(size_t)s * m + j] = th[j];
          for (int i = 0; i < p; ++i) {
            const double v = C[(size_t)i * r + j];
            cs[(size_t)i * qout + j] = v;
            if (with_p && i >= m && !locked[(size_t)s * m + j]) cs[(size_t)i * qout + m + j] = v;  // P = S C, X rows zeroed
          }
        } else {
          lam[(size_t)s * m + j] = 1e300;  // fewer independent directions than m
        }
      }
    };
    on_host_threads(rr_one, s0, s1);   // the per-subdomain projected problems are independent: one host thread each (bounded)
    if (want_prop && last) {
      prop_pending = true;   // hGA / hGB will hold the [X P] blocks of the new basis once run_propagate has run
      have_prop = true;
    }
    t_rr_host += secs(tg1, clk::now());
    const size_t co = (size_t)s0 * p * qout, mo = (size_t)s0 * m;    // the group's offsets in the per-subdomain arrays
    // stream-ordered uploads from pinned memory, no host synchronisation: the host next writes these arrays in the next
    // Rayleigh-Ritz of the same group, which follows a synchronous download behind the kernels that read them
    bk::h2d_async(dC + co, hC.data() + co, sizeof(double) * (size_t)(s1 - s0) * p * qout);
    if (fused_update && p == p3 && with_p) {
      // one launch: [X' P'] for S, A S, B S (the [P W] product once per operand) and the next residual block
      for (size_t e = mo; e < (size_t)s1 * m; ++e) keep[e] = (locked[e] || frozen[e / m]) ? 0.0 : 1.0;
      double* hk = hKL.data() + mo;
      double* hl = hKL.data() + (size_t)ns * m + mo;
      std::copy(keep.begin() + mo, keep.begin() + (size_t)s1 * m, hk);
      std::copy(lam.begin() + mo, lam.begin() + (size_t)s1 * m, hl);
      bk::h2d_async(dkeep + mo, hk, sizeof(double) * (size_t)(s1 - s0) * m);
      bk::h2d_async(dlam + mo, hl, sizeof(double) * (size_t)(s1 - s0) * m);
      if (lean) bk::lobpcg_update32_basis(*cg, S, dC + co, dkeep + mo, T);
      else bk::lobpcg_update32(*cg, S, AS, BS, dC + co, dkeep + mo, dlam + mo, dmask + mo, T, AT, BT, cr);
      if (last) have_R = true;   // lean: nothing left for the host to do -- the device phase starts with the residual product
    } else {
      if (s0 != 0 || s1 != ns) throw std::runtime_error("lobpcg: grouped update without the fused kernel");
      have_R = false;
      bk::block_mul(ch, S, p3, p, dC, qout, T, p3, false);
      bk::block_mul(ch, AS, p3, p, dC, qout, AT, p3, false);
      bk::block_mul(ch, BS, p3, p, dC, qout, BT, p3, false);
    }
    if (last) {
      std::swap(S, T);
      if (!lean) { std::swap(AS, AT); std::swap(BS, BT); }
    }
    return 0;
  };
  gram_blocks(m);
  rayleigh_ritz(m, 0, m, false);
  // P block := 0 (the swap left stale data there)
  bk::block_axpby(S + m, p3, 0.0, S + m, p3, 0.0, nL, m);
  bk::block_axpby(AS + m, p3, 0.0, AS + m, p3, 0.0, nL, m);
  bk::block_axpby(BS + m, p3, 0.0, BS + m, p3, 0.0, nL, m);

  const double tol = P.tol > 0.0 ? P.tol : opt.eps_tol;
  const double lmax = P.lmax * 1.05, lmin = lmax / std::max(1.5, opt.cheb_ratio);
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  // Convergence test (-els2_eps_conv):
  //   "sinvert" (default with the V-cycle preconditioner): ARPACK's test in the shift-invert mode the reference runs
  //   (geneo.cpp:649-663, EPSSetTolerances tol at :658): Ritz estimate || OP x - theta x ||_B <= tol theta with
  //   OP = A^-1 B, theta = 1 / lambda, i.e. || A^-1 (A x - lambda B x) ||_B <= tol ||x||_B.  The preconditioned residual
  //   W = T r (T = one V-cycle ~ A^-1) is the next search direction anyway and its B-norm is on the diagonal of the Gram
  //   block the Rayleigh-Ritz needs: the test costs nothing and no extra host round trip.
  //   "residual": || r ||_2 <= tol (||A x||_2 + |lambda| ||B x||_2) (round 1; kept for the Chebyshev preconditioner,
  //   which is no approximation of A^-1 on the smooth components, and for checks of B with A = identity).
  const bool conv_sinvert = P.amg && opt.eps_conv != "residual";
  conv_sinvert_now = conv_sinvert;
  fused_update = conv_sinvert && m == 32 && bk::lobpcg_update32_available() && !getenv("GENEO_LOBPCG_NO_FUSED_UPDATE");
  static const bool no_lean = getenv("GENEO_LOBPCG_NO_LEAN") != nullptr;
  // (not with GENEO_LOBPCG_FULL_GRAM: explicit 96 x 96 Grams in every iteration read all of A S and B S)
  lean = fused_update && !no_lean && !pipeline_ok && !full_gram && P.dual_pat && bk::spmm_dual_available(*P.dual_pat, m);
  if (getenv("GENEO_DEBUG") && !lean)
    fprintf(stderr, "[lobpcg %s] carried form: shift-invert test %d, block %d, fused update %d, shared pattern %d\n", P.label,
            (int)conv_sinvert, m, (int)fused_update, (int)(P.dual_pat != nullptr));
  if (lean) {   // the second set of product blocks is never written
    for (double* q : {AT, BT}) {
      owned_bufs.erase(std::find(owned_bufs.begin(), owned_bufs.end(), q));
      bk::dfree(q);
    }
    AT = BT = nullptr;
  }
  int it = 0;
  bool all_done = false;
  std::vector<int> nev_s(ns);
  for (int s = 0; s < ns; ++s)
    nev_s[s] = std::max(0, std::min(nev_try, rows_of(s) - (P.locked_cols ? P.locked_cols[s] : 0)));
  static const bool nolock = getenv("GENEO_LOBPCG_NOLOCK") != nullptr;
  std::vector<double> lam_prev((size_t)ns * m, 1e300);   // Ritz values of the previous iteration (straggler test)
  // A pair is ACCEPTED at the tolerance but keeps iterating (its W / P columns stay in the basis) until it is far below
  // it: the reference's ARPACK stops when the slowest wanted pair meets tol, by which time the others are orders of
  // magnitude better; locking every pair AT tol would leave the whole coarse space at the tolerance's worst case (64^3,
  // tol 1e-3: Ritz values 2e-3 off inside the dense cluster, outer PCG 26 instead of 24 iterations).  The lock proper
  // (columns dropped: at tight tolerances they would only inject rounding noise) comes at max(tol^2, min(tol, 1e-10)).
  static const bool lock_at_tol = getenv("GENEO_LOBPCG_LOCK_AT_TOL") != nullptr;   // experiment: round-1 behaviour
  const double tol_lock = lock_at_tol ? tol : std::max(tol * tol, std::min(tol, 1e-10));
  std::vector<char> conv((size_t)ns * m, 0);
  // locks + frozen subdomains from res[][]; returns true when every subdomain is done
  auto update_locks = [&](int s0 = 0, int s1 = -1) {
    bool done = true;
    if (s1 < 0) s1 = ns;
    for (int s = s0; s < s1; ++s) {
      if (frozen[s]) continue;
      bool sub_done = true;
      for (int j = 0; j < m; ++j) {
        const size_t e = (size_t)s * m + j;
        // soft locking: a converged pair stays in X (and in the Rayleigh-Ritz) but no longer
        // contributes search directions -- its W / P columns would only inject rounding noise
        if (lam[e] >= 1e299 || (res[s][j] <= tol_lock && !nolock)) locked[e] = 1;
        if (lam[e] >= 1e299 || res[s][j] <= tol) conv[e] = 1;      // sticky: a pair that met the tolerance stays accepted
        if (j < nev_s[s] && !conv[e]) sub_done = false;
      }
      // No straggler: a Ritz pair behind the wanted ones that has not converged may still be on its way to an
      // eigenvalue BELOW the last wanted one (a late copy of a multiplet: symmetric subdomains have 3- and 6-fold
      // ones) -- the wanted pairs would then all be converged eigenpairs, but not the lowest ones.  Sorted Ritz
      // values only move down from one iteration to the next, so a pair is harmless when either
      //   (a) its shift-invert estimate e_j brackets its eigenvalue above the last wanted one:
      //       theta_j / (1 + e_j) > last (OP = A^-1 B is self-adjoint in the B inner product), or
      //   (b) it has all but stopped moving: even 12 x its last step down (a geometric tail of ratio 0.92, slower
      //       than any convergence seen on this pencil) leaves it above the last wanted one.
      // Pairs inside the cluster of the last wanted value satisfy neither and have to converge themselves.
      if (sub_done && conv_sinvert && !nolock && nev_s[s] >= 1 && nev_s[s] < m) {
        // "above the last wanted one" up to the resolution of the tolerance itself: an eigenvalue within 2 tol
        // (relative) of it is, at this tolerance, the same eigenvalue for the coarse space (ARPACK's own test resolves
        // theta to tol * theta); without the slack the dense band behind the wanted pairs (spacing 0.1 - 0.3 %) keeps
        // one subdomain iterating long after its wanted pairs are an order of magnitude below tol (126^3: 12 of 80)
        static const double slack = getenv("GENEO_LOBPCG_SLACK") ? atof(getenv("GENEO_LOBPCG_SLACK")) : 2.0;
        const double last = lam[(size_t)s * m + nev_s[s] - 1] * (1.0 - slack * tol);
        for (int j = nev_s[s]; j < m; ++j) {
          const size_t e = (size_t)s * m + j;
          if (conv[e] || lam[e] >= 1e299) continue;
          const double drop = (lam_prev[e] < 1e299) ? std::max(0.0, lam_prev[e] - lam[e]) : lam[e];
          if (lam[e] / (1.0 + res[s][j]) > last) continue;
          if (lam[e] - 12.0 * drop > last) continue;
          sub_done = false;
        }
      }
      if (sub_done) frozen[s] = 1;
      else done = false;
    }
    return done;
  };
  auto debug_line = [&]() {
    if (!getenv("GENEO_DEBUG")) return;
    fprintf(stderr, "[lobpcg %s] it %d maxres:", P.label, it);
    for (int s = 0; s < ns; ++s) {
      double mx = 0.0;
      for (int j = 0; j < nev_s[s]; ++j) mx = std::max(mx, res[s][j]);
      fprintf(stderr, " %.2e", mx);
    }
    fprintf(stderr, " | locked");
    for (int s = 0; s < ns; ++s) {
      int nl = 0;
      for (int j = 0; j < m; ++j) nl += locked[(size_t)s * m + j] ? 1 : 0;
      fprintf(stderr, " %d", nl);
    }
    fprintf(stderr, " | prefix");
    for (int s = 0; s < ns; ++s) {
      int k = 0;
      while (k < m && (locked[(size_t)s * m + k] || frozen[s])) ++k;
      fprintf(stderr, " %d", k);
    }
    fprintf(stderr, " | lam0 %.6e %.6e .. %.6e | pc %s\n", lam[0], lam[1], lam[std::max(1, nev_s[0]) - 1], P.amg ? "amg" : "cheb");
  };
  const int max_it = P.max_it > 0 ? P.max_it : opt.eps_max_it;
  for (it = 0; it <= max_it; ++it) {
    // the A X / B X blocks are carried by recurrence (A S C); refresh them explicitly every few
    // iterations so that rounding drift (eps * ||A|| ||x|| per update, large for high-contrast
    // operators) cannot accumulate into the residual
    static const int refresh = getenv("GENEO_LOBPCG_REFRESH") ? atoi(getenv("GENEO_LOBPCG_REFRESH")) : 8;
    const bool refreshed = (it > 0 && refresh > 0 && it % refresh == 0);
    if (refreshed && lean) {   // the explicit Gram blocks of this iteration need A [X P] and B [X P]
      bk::spmm_dual(*P.dual_pat, P.dual_vA, P.dual_vB, S, p3, AS, BS, p3, m);
      bk::spmm_dual(*P.dual_pat, P.dual_vA, P.dual_vB, S + m, p3, AS + m, BS + m, p3, m);
      info.eig_spmm += 4;
    } else if (refreshed) {
      applyA(S, AS);
      applyB(S, BS);
    }
    // residual into the W slot, convergence test (the fused update has already uploaded the Ritz values it used)
    if (!have_R || (refreshed && !lean)) bk::h2d(dlam, lam.data(), sizeof(double) * (size_t)ns * m);
    double* W = S + 2 * m;
    // The device part of one iteration -- residual, preconditioner, A W, B W, the two Gram blocks: ~45 launches, many of
    // them on the small coarse levels of the V-cycle where the host cannot issue as fast as the GPU retires.  With the
    // shift-invert test nothing in it needs the host, so it is captured once per buffer parity (S / T swap roles every
    // iteration) into a HIP graph and replayed.
    auto precondition = [&]() {
      if (P.amg) {
        // W = T r : one smoothed-aggregation V-cycle of A on the whole block (~ shift-invert at sigma = 0)
        if (P.amg_level > 0) P.amg->vcycle_from(P.amg_level, cr, m, W, p3, m);
        else P.amg->vcycle(cr, m, W, p3, m);
        info.eig_spmm += 3;
      } else {
        // W = T r : Chebyshev iteration on A z = r with Jacobi scaling, z0 = 0 (Saad, Alg. 12.1)
        double rho = 1.0 / sigma;
        bk::block_rowscale(cd, m, cr, m, P.dinv, 1.0 / theta, 0.0, nL, m);  // d = Dinv r / theta
        bk::block_axpby(W, p3, 1.0, cd, m, 0.0, nL, m);                      // z = d
        for (int k = 1; k < opt.cheb_degree; ++k) {
          bk::spmm_strided(*P.A, cd, m, cad, m, m, P.As, P.As);              // A d
          info.eig_spmm++;
          const double rho_new = 1.0 / (2.0 * sigma - rho);
          // r -= A d ; d = (2 rho'/delta) Dinv r + rho' rho d ; z += d   (one fused pass)
          bk::cheb_update(cr, cad, cd, W, p3, P.dinv, 2.0 * rho_new / delta, rho_new * rho, nL, m);
          rho = rho_new;
        }
      }
      deflate(W, p3);
    };
    const bool reduced = have_prop && !full_gram && !(refresh > 0 && it % refresh == 0);
    // Two-stage pipeline of the host part (pipe): the subdomains are split into two groups; each group has its own Gram
    // launches, Rayleigh-Ritz and update launch.  While the host solves the projected problems of group 0 the GPU
    // computes the Gram rows of group 1, and while it solves those of group 1 the GPU updates the basis of group 0:
    // of the ~1.3 ms per iteration the GPU used to wait for the host, ~0.3 ms are left.
    const bool pipe = pipeline_ok && reduced && fused_update && conv_sinvert;
    auto device_phase = [&](bool with_gram) {
      // residual block (columns locked in EARLIER iterations come out zero): already written by the fused update,
      // unless A X / B X have just been refreshed
      if (lean) {
        bk::spmm_dual_residual(*P.dual_pat, P.dual_vA, P.dual_vB, S, p3, cr, m, m, ch, dlam, dmask);
        info.eig_spmm += 2;
      } else if (!have_R || refreshed) {
        bk::block_residual_norms(ch, AS, p3, BS, p3, dlam, m, cr, m, dmask, nullptr);
      }
      precondition();
      if (P.dual_pat && bk::spmm_dual_available(*P.dual_pat, m)) {    // A W and B W in one pass over W
        bk::spmm_dual(*P.dual_pat, P.dual_vA, P.dual_vB, W, p3, AS + 2 * m, BS + 2 * m, p3, m);
        info.eig_spmm += 2;
      } else {
        applyA(W, AS + 2 * m);
        applyB(W, BS + 2 * m);
      }
      if (!with_gram) return;
      if (reduced) {       // (A W)^T [X P W] and (B W)^T [X P W]: the W rows of the two Gram matrices, one pass over S
        bk::gram2(ch, AS + 2 * m, p3, m, BS + 2 * m, p3, m, S, p3, p3, dGW);
      } else {
        bk::gram(ch, S, p3, p3, AS, p3, p3, dGA);
        bk::gram(ch, S, p3, p3, BS, p3, p3, dGB);
      }
    };
    if (!conv_sinvert) {
      bk::block_residual_norms(ch, AS, p3, BS, p3, dlam, m, cr, m, dmask, dn3);
      bk::d2h(n3.data(), dn3, sizeof(double) * (size_t)ns * 3 * m);
      for (int s = 0; s < ns; ++s)
        for (int j = 0; j < m; ++j) {
          const size_t e = (size_t)s * m + j;
          const double nrj = n3[(size_t)s * 3 * m + j], naj = n3[(size_t)s * 3 * m + m + j], nbj = n3[(size_t)s * 3 * m + 2 * m + j];
          const double den = std::sqrt(naj) + std::fabs(lam[e]) * std::sqrt(nbj);
          res[s][j] = den > 0 ? std::sqrt(nrj) / den : 0.0;
        }
      all_done = update_locks();
      debug_line();
      if (all_done || it == max_it) break;
      precondition();
      applyA(W, AS + 2 * m);
      applyB(W, BS + 2 * m);
      gram_blocks(p3);
      bool changed = false;
      for (size_t e = 0; e < locked.size(); ++e) {
        const double mk = (locked[e] || frozen[e / m]) ? 0.0 : 1.0;
        if (mk != mask[e]) changed = true;
        mask[e] = mk;
      }
      if (changed) bk::h2d(dmask, mask.data(), sizeof(double) * mask.size());
      lam_prev = lam;
      rayleigh_ritz(p3, m, 2 * m, true);
      continue;
    }
    // it 0 runs direct (first calls size scratch buffers); while bench.py's in-situ kernel timer is on every 8th
    // iteration (and the 4th: a solve that started from coarse Ritz vectors has only a handful) runs direct so that its
    // launches can be bracketed by events (graph nodes cannot)
    const int par = it & 1;
    const bool direct = no_graph || !reduced || (fused_update && !have_R) || (bk::spmv_profiling() && (it % 8 == 1 || it == 4));
    if (!direct && !it_graph[par] && !it_graph_failed) {
      const int spmm_before = info.eig_spmm;
      if (bk::graph_capture_begin()) {
        try {
          device_phase(!pipe);
        } catch (...) {
          bk::graph_capture_end();
          throw;
        }
        it_graph[par] = bk::graph_capture_end();
        graph_has_gram[par] = !pipe;
      }
      info.eig_spmm = spmm_before;   // recorded, not run
      if (!it_graph[par]) it_graph_failed = true;
    }
    if (!direct && it_graph[par] && graph_has_gram[par] == !pipe) {
      bk::graph_launch(it_graph[par]);
      info.eig_spmm += (P.amg ? 5 : 2 + opt.cheb_degree - 1) + (lean ? 2 : 0);   // what device_phase counts when it runs direct
    } else {
      device_phase(!pipe);
    }
    const int ngr = pipe ? 2 : 1;
    if (pipe)
      for (int g = 0; g < 2; ++g) {      // the Gram rows of each group, an event behind each pair of launches
        bk::gram2(chg[g], AS + 2 * m, p3, m, BS + 2 * m, p3, m, S, p3, p3, dGW + (size_t)grp[g] * 2 * m * p3);
        bk::event_record(gram_ev[g]);
      }
    if (reduced) run_propagate();          // host work hidden behind the launches above
    else prop_pending = false;             // the explicit 96 x 96 blocks are on their way
    bool all = true, stop = false;
    for (int g = 0; g < ngr; ++g) {
      const int s0 = pipe ? grp[g] : 0, s1 = pipe ? grp[g + 1] : ns;
      auto tg0 = clk::now();
      if (reduced) {
        const size_t go = (size_t)s0 * 2 * m * p3, gn = (size_t)(s1 - s0) * 2 * m * p3;
        if (pipe) bk::d2h_after(hGw.data() + go, dGW + go, sizeof(double) * gn, gram_ev[g]);
        else bk::d2h(hGw.data() + go, dGW + go, sizeof(double) * gn);
        for (int which = 0; which < 2; ++which) {
          std::vector<double>& G = which ? hGB : hGA;
          for (int sd = s0; sd < s1; ++sd) {
            double* gm = G.data() + (size_t)sd * p3 * p3;
            const double* gw = hGw.data() + ((size_t)sd * 2 + which) * m * p3;
            for (int a = 0; a < m; ++a)
              for (int b = 0; b < p3; ++b) {
                const double v = gw[(size_t)a * p3 + b];
                gm[(size_t)(2 * m + a) * p3 + b] = v;
                if (b < 2 * m) gm[(size_t)b * p3 + 2 * m + a] = v;
              }
          }
        }
      } else {
        bk::d2h(hGA.data(), dGA, sizeof(double) * (size_t)ns * p3 * p3);
        bk::d2h(hGB.data(), dGB, sizeof(double) * (size_t)ns * p3 * p3);
      }
      t_dev_wait += secs(tg0, clk::now());
      // || T r_j ||_B / || x_j ||_B off the diagonal of S^T B S (columns masked earlier have W_j = 0: they stay locked)
      for (int s = s0; s < s1; ++s)
        for (int j = 0; j < m; ++j) {
          const double* gb = hGB.data() + (size_t)s * p3 * p3;
          const double ww = gb[(size_t)(2 * m + j) * p3 + 2 * m + j], xx = gb[(size_t)j * p3 + j];
          res[s][j] = (xx > 0.0 && ww > 0.0) ? std::sqrt(ww / xx) : 0.0;
        }
      all = update_locks(s0, s1) && all;
      if (g == ngr - 1) {
        // (with two groups the first one has been updated already when the last one turns out to be the last to
        // converge: its new blocks went to the T buffers, which are simply not swapped in; a group that still had
        // an unconverged subdomain rules `all` out, so no Ritz value of a finished run is ever overwritten)
        all_done = all;
        debug_line();
        if (all_done || it == max_it) { stop = true; break; }
      }
      // soft locking without extra passes: the P columns of locked pairs are dropped through the Rayleigh-Ritz
      // coefficients (rr_one) and their residual columns through the mask of the next block_residual_norms
      bool changed = false;
      for (size_t e = (size_t)s0 * m; e < (size_t)s1 * m; ++e) {
        const double mk = (locked[e] || frozen[e / m]) ? 0.0 : 1.0;
        if (mk != mask[e]) changed = true;
        mask[e] = mk;
      }
      if (changed) bk::h2d(dmask + (size_t)s0 * m, mask.data() + (size_t)s0 * m, sizeof(double) * (size_t)(s1 - s0) * m);
      std::copy(lam.begin() + (size_t)s0 * m, lam.begin() + (size_t)s1 * m, lam_prev.begin() + (size_t)s0 * m);
      rayleigh_ritz(p3, m, 2 * m, true, s0, s1, pipe ? &chg[g] : &ch, g == ngr - 1);
    }
    if (stop) break;
  }
  if (P.iterations) *P.iterations += it;
  else info.eig_iterations += it;
  if (all_done || P.max_it > 0) bk::block_axpby(Xc, m, 1.0, S, p3, 0.0, nL, m);
  bk::sync();
  if (getenv("GENEO_DEBUG"))
    fprintf(stderr, "[lobpcg %s] %d iterations (%s) %.3f s: host Rayleigh-Ritz %.3f s (+ %.3f s of Gram propagation behind the device phase), waiting for the Gram blocks %.3f s\n", P.label, it,
            lean ? "lean: S alone carried" : "A S, B S carried", secs(t_lob0, clk::now()), t_rr_host, t_prop_host, t_dev_wait);
  cleanup();
  if (!all_done && P.max_it <= 0) {
    // The reference aborts on EPS_DIVERGED_ITS (checkEPSSolve, geneo.cpp:577-624).
    std::ostringstream msg;
    msg << "GenEO preconditioner: els2-" << P.label << " KO (EPS_DIVERGED_ITS after " << it << " LOBPCG iterations)";
    return fail(msg.str());
  }
  return 0;
}

// Number of eigenpairs asked per problem (geneo.cpp:855-879 without the inertia count: the block
// always carries the -geneo_cut cap, the threshold filter removes the rest) and the LOBPCG block.
int PC::eig_targets(int* nev_try) const {
  int cut = opt.cut;
  if (opt.lvl2 == 2 && cut >= 2) cut /= 2;  // GenEO-2 has two eigenproblems, geneo.cpp:1275
  int nev = cut > 0 ? cut : opt.eps_nev;
  if (opt.noSyl && cut > 0) nev = std::min(opt.eps_nev, cut);  // geneo.cpp:855,:871-879
  if (opt.noSyl && cut <= 0) nev = opt.eps_nev;
  if (nev_try) *nev_try = nev;
  if (opt.eps_block > 0) return opt.eps_block;
  const int want = nev + std::max(4, nev / 4);
  return want <= 16 ? 16 : (want <= 32 ? 32 : 64);
}

// widest block the eigensolver may use (sizes the V-cycle work space): without -geneo_cut the block can grow to 64
int PC::eig_block_max() const {
  if (opt.cut <= 0 && !opt.noSyl && opt.eps_block <= 0) return 64;
  return eig_targets(nullptr);
}

// getLocalGenEOTau, geneo.cpp:1097-1118
void PC::local_tau() {
  const int ns = (int)subs.size();
  tauLoc.assign(ns, opt.tau);
  if (opt.cst) return;
  for (int s = 0; s < ns; ++s) {
    int k = 0;
    for (int v : subs[s].mult) k = std::max(k, v);
    double t = k * opt.tau;
    if (t >= 1.0) t = 0.9;
    tauLoc[s] = t;
  }
}

// getLocalGenEOGamma, geneo.cpp:1120-1232 -- follows the code, including the test at :1143-1145 that
// stores 0 where two subdomains DO intersect.
int PC::local_gamma() {
  const int ns = (int)subs.size(), Pn = nsub_global;
  gammaLoc.assign(ns, opt.gamma);
  if (opt.cst) return 0;
  // emptiness flags of intersectLoc: supplied (PCGenEOSetIntersect) or, on one rank, derived from the maps
  std::vector<double> c((size_t)Pn * Pn, 0.0);
  bool have_all = true;
  for (auto& s : subs) have_all = have_all && (int)s.intersect.size() == Pn;
  if (!have_all) {
    if (size > 1) return fail("GenEO-2: intersection flags are required on several ranks (PCGenEOSetIntersect), or use -geneo_cst");
    std::vector<std::vector<int>> owners(N);
    for (int s = 0; s < ns; ++s)
      for (int g : subs[s].l2g) owners[g].push_back(s);
    for (auto& s : subs) s.intersect.assign(Pn, 0);
    for (auto& o : owners)
      for (int a : o)
        for (int b : o)
          if (a != b) subs[a].intersect[subs[b].gid] = 1;
  }
  for (auto& s : subs)
    for (int q = 0; q < Pn; ++q) c[(size_t)s.gid * Pn + q] = (q == s.gid) ? 1.0 : (s.intersect[q] ? 0.0 : 1.0);
  if (size > 1) {  // MPI all_gather of the rows (geneo.cpp:1151-1160) through one all-reduce
    double* dc = (double*)bk::alloc(sizeof(double) * c.size());
    bk::h2d(dc, c.data(), sizeof(double) * c.size());
    allreduce(dc, (int)c.size());
    bk::d2h(c.data(), dc, sizeof(double) * c.size());
    bk::dfree(dc);
  }
  std::vector<double> f(Pn), mth((size_t)Pn * Pn);
  for (int r = 0; r < Pn; ++r) {
    double sum = 0.0;
    for (int q = 0; q < Pn; ++q) sum += c[(size_t)r * Pn + q];
    f[r] = 1.0 / sum;
  }
  for (int r = 0; r < Pn; ++r)
    for (int q = 0; q < Pn; ++q)
      mth[(size_t)r * Pn + q] = 0.5 * (c[(size_t)r * Pn + q] + c[(size_t)q * Pn + r]) * f[r] * f[q];
  std::vector<double> w, V;
  dense::sym_eig(mth, Pn, w, V);
  double lmax = w[0];
  for (double v : w)
    if (std::fabs(v) > std::fabs(lmax)) lmax = v;
  for (int s = 0; s < ns; ++s) {
    double g = opt.gamma / lmax;
    g = g * f[subs[s].gid] * f[subs[s].gid];
    if (g <= 1.0) g = 1.1;
    gammaLoc[s] = g;
  }
  return 0;
}

// GenEO-1: A_Neu v = lambda (D A_Dir D) v, lambda <= tau.   GenEO-2 (geneo.cpp:1274-1300): A_Neu v = lambda A_Rob v,
// lambda <= tau_loc, then (D A_Dir D) v = lambda A_Rob v, lambda >= gamma_loc -- the LARGEST eigenvalues, computed
// as the lowest mu = 1/lambda of A_Rob v = mu (D A_Dir D) v with the V-cycle of A_Rob as preconditioner.
int PC::eigen_lobpcg() {
  const int ns = (int)subs.size();
  const bool g2 = (opt.lvl2 == 2);
  int nev_try = 0;
  const int m = eig_targets(&nev_try);
  if (m != 16 && m != 32 && m != 64) return fail("GenEO: -els2_eps_block must be 16, 32 or 64");
  if (nev_try > m) return fail("GenEO: -geneo_cut / -els2_eps_nev larger than the LOBPCG block (max 64)");
  auto dv = [](size_t n) { return (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, n)); };
  // B = D A_Dir D always uses the true Dirichlet matrix: rebuild if level 1 holds the Robin one
  bk::Csr dirB = dirL;
  bool own_dirB = false;
  if (opt.lvl1ORAS) {
    std::vector<const HostCsr*> dm(ns);
    for (int s = 0; s < ns; ++s) dm[s] = &subs[s].a_dir;
    dirB = upload_blockdiag(dm, suboff, nullptr);
    dirB.fine = true;
    own_dirB = true;
  }
  // B = D A_Dir D (geneo.cpp:1243-1247, MatDiagonalScale on a copy) as a values-only copy sharing A_Dir's index
  // arrays: the block products with B then run without the two scaling gathers per entry
  bk::Csr dirBD = bk::csr_scaled_alias(dirB, d_D, d_D, false);
  // GenEO-1: the pattern of A_Dir contains that of A_Neu (A_Dir = R A R^T holds every coupling among the subdomain's
  // nodes, A_Neu those of the subdomain's own elements), so A_Neu can ride on A_Dir's sliced layout next to D A_Dir D and
  // LOBPCG computes A W and B W in ONE pass over W (bk::spmm_dual).  nullptr (not contained / not on the sliced path):
  // two products as before.
  double* neu_on_dir = (!g2 && !getenv("GENEO_LOBPCG_NO_DUAL")) ? bk::sell_values_on(dirB, neuL) : nullptr;
  auto release = [&]() {
    if (neu_on_dir) bk::dfree(neu_on_dir);
    neu_on_dir = nullptr;
    bk::csr_free(dirBD);
    if (own_dirB) bk::csr_free(dirB);
  };
  info.eig_iterations = 0;
  std::vector<HostCsr> robh;
  std::vector<const HostCsr*> hostB(ns);
  if (opt.check) {   // host copies of the pencils' right-hand matrices for checkSPD
    if (g2) robh.resize(ns);
    for (int s = 0; s < ns; ++s) {
      if (g2) {
        make_robin(subs[s], robh[s]);
        hostB[s] = &robh[s];
      } else {
        hostB[s] = &subs[s].a_dir;
      }
    }
  }
  // Without -geneo_cut the reference keeps EVERY eigenvalue on the wanted side of the threshold (its nev is the LDLt
  // inertia count, geneo.cpp:502-560, capped only by n and -geneo_cut; :713 keeps every lambda <= tau).  There is no
  // inertia count here.  The block is doubled (16 -> 32 -> 64 columns) and the problem solved again while the whole
  // block lands on the wanted side; once 48 pairs of a 64-column block are all wanted, they are LOCKED and the
  // iteration restarts with a fresh block in the B-orthogonal complement of everything locked so far (deflated
  // restart: lobpcg_solve projects the start block and every preconditioned residual), 48 more pairs per stage, until
  // the threshold falls inside a block in every subdomain.  A subdomain whose threshold has been found sits out the
  // later stages (frozen from the start).  The only ceiling left is the 256 coarse vectors per subdomain of the
  // coarse-space kernels (ZMAXK), reported as an error.
  struct Stage {
    double* X = nullptr;            // n_L x m Ritz vectors of this stage
    int m = 0;
    std::vector<double> lam;        // ns x m
    std::vector<int> count;         // per subdomain: leading columns that are candidates (0: it sat this stage out)
  };
  std::vector<double*> lock_bufs;   // Y / B Y copies of the locked blocks
  auto free_stages = [&](std::vector<Stage>& st) {
    for (auto& g : st)
      if (g.X) bk::dfree(g.X);
    st.clear();
  };
  auto solve_grow = [&](EigProblem P, bool gamma, std::vector<Stage>& out) -> int {
    int mm = m, nev = nev_try;
    auto wanted = [&](int s, double l) { return gamma ? (l > 0.0 && 1.0 / l >= gammaLoc[s]) : (l <= tauLoc[s]); };
    Stage cur;
    for (;;) {   // growth by re-solving with a wider block (cheap: these blocks are small)
      P.nev_try = nev;
      cur.X = dv((size_t)nL * mm);
      cur.m = mm;
      if (int rc = lobpcg_solve(P, mm, cur.lam, cur.X)) { bk::dfree(cur.X); return rc; }
      P.X0 = nullptr;                 // a start block belongs to the first solve (its width)
      cur.count.assign(ns, 0);
      for (int s = 0; s < ns; ++s) cur.count[s] = std::min(nev, (int)subs[s].l2g.size());
      if (opt.cut > 0 || opt.noSyl || opt.eps_block > 0) { out.push_back(cur); return 0; }
      bool full = false;
      for (int s = 0; s < ns; ++s) {
        const int n_s = (int)subs[s].l2g.size(), nv = cur.count[s];
        if (nv >= n_s) continue;
        const double last = cur.lam[(size_t)s * mm + nv - 1];
        if (last < 1e299 && wanted(s, last)) full = true;
      }
      if (!full) { out.push_back(cur); return 0; }
      if (nev >= 48) break;
      bk::dfree(cur.X);
      cur.X = nullptr;
      nev = std::min(48, 2 * nev);
      const int want = nev + std::max(4, nev / 4);
      mm = want <= 16 ? 16 : (want <= 32 ? 32 : 64);
    }
    // ---- deflated restarts: 64-column blocks, 48 pairs asked per stage
    // Locked vectors define the space every later stage works in: an error of size tol in them comes back at first
    // order in the vectors of the later stages.  The stages therefore run at min(-els2_eps_tol, 1e-8), the first one
    // again from its start block (with a loose -els2_eps_tol the reference's ARPACK, asked for ALL these pairs at
    // once, returns most of them far below the tolerance as well; round 2 handed such subdomains to a dense solver).
    const double tight = std::min(opt.eps_tol, 1e-8);
    if (tight < opt.eps_tol) {
      P.tol = tight;
      P.nev_try = nev;
      if (int rc = lobpcg_solve(P, mm, cur.lam, cur.X)) { bk::dfree(cur.X); return rc; }
    }
    std::vector<EigProblem::Locked> locked_blocks;
    std::vector<int> locked_cols(ns, 0);
    std::vector<char> skip(ns, 0);
    for (int stage = 1;; ++stage) {
      // lock the candidates of the subdomains that are still full; the others are done
      std::vector<double> cs((size_t)ns * cur.m, 0.0);
      bool any = false;
      for (int s = 0; s < ns; ++s) {
        const int n_s = (int)subs[s].l2g.size(), nv = cur.count[s];
        bool full = !skip[s] && nv > 0 && locked_cols[s] + nv < n_s;
        if (full) {
          const double last = cur.lam[(size_t)s * cur.m + nv - 1];
          full = last < 1e299 && wanted(s, last);
        }
        if (!full) { skip[s] = 1; continue; }
        any = true;
        for (int j = 0; j < nv; ++j) cs[(size_t)s * cur.m + j] = 1.0;
        locked_cols[s] += nv;
        if (locked_cols[s] + 48 > 256) {
          out.push_back(cur);      // the stage's vectors belong to `out` from here on: release_all() frees them
          return fail("GenEO: more than 256 eigenvalues pass the threshold in one subdomain (coarse-space kernels hold 256 "
                      "vectors per subdomain): set -geneo_cut or lower -geneo_tau");
        }
      }
      out.push_back(cur);
      if (!any) return 0;
      double* Y = dv((size_t)nL * cur.m);
      double* BY = dv((size_t)nL * cur.m);
      lock_bufs.push_back(Y);
      lock_bufs.push_back(BY);
      {
        struct DevBuf { double* p; ~DevBuf() { bk::dfree(p); } } dcs{dv(cs.size())};   // released on a throwing launch too
        bk::h2d(dcs.p, cs.data(), sizeof(double) * cs.size());
        bk::block_axpby(Y, cur.m, 1.0, cur.X, cur.m, 0.0, nL, cur.m);
        bk::block_colscale(ch, Y, cur.m, cur.m, dcs.p);
        bk::spmm_strided(*P.B, Y, cur.m, BY, cur.m, cur.m, P.Bs, P.Bs);
        bk::sync();
      }
      locked_blocks.push_back({Y, BY, cur.m});
      Stage nxt;
      nxt.m = 64;
      nxt.X = dv((size_t)nL * 64);
      EigProblem Pd = P;
      Pd.nev_try = 48;
      Pd.defl = &locked_blocks;
      Pd.skip = skip.data();
      Pd.locked_cols = locked_cols.data();
      Pd.seed_off = stage;
      if (int rc = lobpcg_solve(Pd, 64, nxt.lam, nxt.X)) { bk::dfree(nxt.X); return rc; }
      nxt.count.assign(ns, 0);
      for (int s = 0; s < ns; ++s)
        if (!skip[s]) nxt.count[s] = std::max(0, std::min(48, (int)subs[s].l2g.size() - locked_cols[s]));
      cur = nxt;
    }
  };
  // Coarse start (Options::eig_coarse_start): Ritz vectors of the Galerkin pencil of multigrid level 1, prolonged -- that
  // solve itself started from level 2, and so on down while a level still holds enough rows for the block (nested
  // iteration).  nullptr: does not apply here (GenEO-2, no hierarchy, small subdomains, the block-growing path without
  // -geneo_cut) or a coarse solve did not deliver m independent pairs per subdomain -- the fine solve then starts from
  // its seeded random block as before.
  info.eig_coarse_iterations = 0;
  // Start block of level l (rows of that level x m, device; the caller frees it) from the pencil of level l + 1:
  // B_{l+1} = R_l Bl P_l, LOBPCG on (A_{l+1}, B_{l+1}) -- started the same way from level l + 2 -- and the prolonged Ritz
  // vectors.  nullptr: level l + 1 is too small for the block, a product exceeded the kernels' capacity, or the solve did
  // not deliver m independent pairs per subdomain.  Bl is borrowed.  Everything made here is released on every way out.
  struct DevBlock {
    double* p = nullptr;
    ~DevBlock() { if (p) bk::dfree(p); }
    double* release() { double* q = p; p = nullptr; return q; }
  };
  struct CsrHold {
    bk::Csr c;
    bool live = false;
    void drop() { if (live) bk::csr_free(c); live = false; }
    ~CsrHold() { drop(); }
  };
  struct ChunksHold {
    bk::Chunks c;
    ~ChunksHold() { if (c.start) bk::chunks_free(c); }
  };
  std::function<double*(int, const bk::Csr&)> start_block = [&](int l, const bk::Csr& Bl) -> double* {
    if (l + 1 >= amgN->nlevels()) return nullptr;
    // (read per set-up: tests lower the row bound to reach the nested form on small grids)
    const int min_rows = getenv("GENEO_COARSE_START_MIN_ROWS") ? atoi(getenv("GENEO_COARSE_START_MIN_ROWS")) : 4096;
    const double ctol = getenv("GENEO_COARSE_START_TOL") ? atof(getenv("GENEO_COARSE_START_TOL")) : 1e-2;
    const std::vector<int>& co = amgN->level_suboff(l + 1);
    const int nc = amgN->level_rows(l + 1);
    // the level must carry the block comfortably in EVERY subdomain of the batch (below level 1: min_rows of them), so that
    // the choice does not depend on which subdomains share a batch as long as all of them qualify
    bool fits = (int)co.size() == ns + 1;
    for (int s = 0; fits && s < ns; ++s) fits = co[s + 1] - co[s] >= std::max(4 * m, l == 0 ? 0 : min_rows);
    if (!fits) return nullptr;
    const auto t0 = clk::now();
    CsrHold BP, Bc;
    bool ok = true;
    BP.c = bk::spgemm(Bl, amgN->level_P(l), nc, &ok);
    BP.live = ok;
    if (!ok) return nullptr;
    Bc.c = bk::spgemm(amgN->level_R(l), BP.c, nc, &ok);
    Bc.live = ok;
    BP.drop();
    if (!ok) return nullptr;
    bk::csr_finish(Bc.c);
    DevBlock below, Xc, X0;
    below.p = start_block(l + 1, Bc.c);
    const auto t1 = clk::now();
    ChunksHold cc;
    cc.c = bk::chunks_upload(ns, co.data());
    EigProblem pc{&amgN->level_A(l + 1), nullptr, &Bc.c, nullptr, amgN, amgN->level_dinv(l + 1), cheb_lmax, nev_try, "tau-coarse"};
    pc.rows = nc;
    pc.chunks = &cc.c;
    pc.row_off = co.data();
    pc.amg_level = l + 1;
    pc.tol = std::max(opt.eps_tol, ctol);     // a start block: the discretisation gap between two levels is larger than this
    pc.iterations = &info.eig_coarse_iterations;
    pc.max_it = 40;
    pc.X0 = below.p;
    std::vector<double> lamc;
    Xc.p = dv((size_t)nc * m);
    const int before = info.eig_coarse_iterations;
    const int rc = lobpcg_solve(pc, m, lamc, Xc.p);
    if (rc) last_error.clear();               // the coarse attempt is optional: its failure is not the set-up's
    bool full = rc == 0;
    for (size_t e = 0; full && e < lamc.size(); ++e) full = lamc[e] < 1e299;
    if (full) {
      X0.p = dv((size_t)amgN->level_rows(l) * m);
      bk::spmm_strided(amgN->level_P(l), Xc.p, m, X0.p, m, m, nullptr, nullptr);
    }
    if (getenv("GENEO_DEBUG")) {
      bk::sync();
      fprintf(stderr, "[coarse start] level %d: %d rows, B products + levels below %.3f s, %d LOBPCG iterations + prolongation %.3f s%s\n",
              l + 1, nc, secs(t0, t1), info.eig_coarse_iterations - before, secs(t1, clk::now()), full ? "" : " -- not used");
    }
    return X0.release();
  };
  auto coarse_start = [&]() -> double* {
    if (g2 || !amgN || opt.els2_pc != "amg" || opt.eig_coarse_start <= 0 || ns == 0) return nullptr;
    if (!(opt.cut > 0 || opt.noSyl || opt.eps_block > 0)) return nullptr;
    if (amgN->level_rows(0) != nL) return nullptr;
    for (int s = 0; s < ns; ++s)            // every subdomain of the batch at the threshold or above
      if ((int64_t)subs[s].l2g.size() < (int64_t)opt.eig_coarse_start) return nullptr;
    return start_block(0, dirBD);
  };
  std::vector<Stage> stT, stG;
  auto release_all = [&]() {
    free_stages(stT);
    free_stages(stG);
    for (double* q : lock_bufs) bk::dfree(q);
    lock_bufs.clear();
    release();
  };
  {
    EigProblem pt{&neuL, nullptr, g2 ? &dirL : &dirBD, nullptr, (opt.els2_pc == "amg") ? amgN : nullptr,
                  d_dinvN, cheb_lmax, nev_try, "tau"};
    if (neu_on_dir && dirBD.sl_val) {
      pt.dual_pat = &dirB;
      pt.dual_vA = neu_on_dir;
      pt.dual_vB = dirBD.sl_val;
    }
    if (opt.check)
      if (int rc = check_local_spd(pt, hostB.data(), !g2)) { release_all(); return rc; }
    double* X0 = coarse_start();
    pt.X0 = X0;
    const int rc_tau = solve_grow(pt, false, stT);
    if (X0) bk::dfree(X0);
    if (rc_tau) { release_all(); return rc_tau; }
  }
  if (g2) {
    if (int rc = local_gamma()) { release_all(); return rc; }
    if (int rc = finish_amg1()) { release_all(); return rc; }   // the gamma problem runs through the level-1 hierarchy
    EigProblem pg{&dirL, nullptr, &dirBD, nullptr, (opt.els2_pc == "amg" && opt.dls1_pc == "amg") ? amg1 : nullptr,
                  d_dinv1, cheb_lmax1, nev_try, "gamma"};
    if (opt.check) {  // the reference checks the B of the gamma pencil as given to SLEPc: A_Rob (geneo.cpp:1299,:884)
      EigProblem pchk = pg;
      pchk.B = &dirL;
      pchk.Bs = nullptr;
      if (int rc = check_local_spd(pchk, hostB.data(), false)) { release_all(); return rc; }
    }
    if (int rc = solve_grow(pg, true, stG)) { release_all(); return rc; }
  }
  for (double* q : lock_bufs) bk::dfree(q);     // the locked copies are no longer needed (the stages keep their own X)
  lock_bufs.clear();
  // ---- selection (geneo.cpp:709-722), Nicolaides (:897-944), empty-Z rule (:1305-1314)
  // sel[stage][s * m + j]: column of that stage's X that becomes the j-th vector the stage contributes to Z_s (-1: the
  // constant vector, carried by the first tau stage)
  auto make_sel = [&](const std::vector<Stage>& st) {
    std::vector<std::vector<int>> v(st.size());
    for (size_t g = 0; g < st.size(); ++g) v[g].assign((size_t)ns * (st[g].m + 1), 0);
    return v;
  };
  std::vector<std::vector<int>> selT = make_sel(stT), selG = make_sel(stG);
  std::vector<std::vector<int>> kT(stT.size(), std::vector<int>(ns, 0)), kG(stG.size(), std::vector<int>(ns, 0));
  std::vector<std::vector<double>> gscale(stG.size());
  for (size_t g = 0; g < stG.size(); ++g) gscale[g].assign((size_t)ns * stG[g].m, 1.0);
  ksub.assign(ns, 0);
  double* ones = d_cg_p;
  double* tmp = d_cg_q;
  bk::set(ones, 1.0, nL);
  bk::spmv(neuL, ones, tmp);
  bk::seg_dot(ch, tmp, ones, d_cg_sc, 8, 0);
  if (g2) {
    bk::spmv(dirL, ones, tmp);
  } else {
    bk::xmy(d_cg_r, ones, d_D, nL);
    bk::spmv(dirB, d_cg_r, tmp);
    bk::xmy(tmp, tmp, d_D, nL);
  }
  bk::seg_dot(ch, tmp, ones, d_cg_sc, 8, 1);
  std::vector<double> sc((size_t)8 * ns);
  bk::d2h(sc.data(), d_cg_sc, sizeof(double) * 8 * ns);
  for (int s = 0; s < ns; ++s) {
    int cnt = 0;
    double minval = 1e300;
    for (size_t g = 0; g < stT.size(); ++g) {
      const Stage& st = stT[g];
      const int stride = st.m + 1;
      for (int j = 0; j < st.count[s]; ++j) {
        const double l = st.lam[(size_t)s * st.m + j];
        if (l >= 1e299) continue;
        candidates[s].push_back(l);
        if (l > tauLoc[s]) continue;
        selT[g][(size_t)s * stride + kT[g][s]++] = j;
        ++cnt;
        eigvals[s].push_back(l);
        minval = std::min(minval, l);
      }
    }
    if (cnt > 0 && minval >= opt.nicolaides_zero * DBL_EPSILON) {
      const double ratio = std::fabs(sc[(size_t)s * 8 + 0] / sc[(size_t)s * 8 + 1]);
      if (ratio <= FLT_EPSILON) {
        const size_t g = stT.size() - 1;      // appended behind every eigenvector, as the reference does (:897-944)
        selT[g][(size_t)s * (stT[g].m + 1) + kT[g][s]++] = -1;
        ++cnt;
        eigvals[s].push_back(0.0);
        info.nicolaidesLoc++;
      }
    }
    int cg = 0;
    for (size_t g = 0; g < stG.size(); ++g) {
      const Stage& st = stG[g];
      const int stride = st.m + 1;
      for (int j = 0; j < st.count[s]; ++j) {
        const double mu = st.lam[(size_t)s * st.m + j];
        if (mu >= 1e299 || !(mu > 0.0)) continue;
        const double l = 1.0 / mu;
        candidates[s].push_back(l);
        if (l < gammaLoc[s]) continue;
        gscale[g][(size_t)s * st.m + j] = std::sqrt(l);  // SLEPc normalisation v^T A_Rob v = 1 (ours: v^T B_w v = 1)
        selG[g][(size_t)s * stride + kG[g][s]++] = j;
        ++cg;
        eigvals[s].push_back(l);
      }
    }
    if (cnt + cg == 0) {
      selT[0][(size_t)s * (stT[0].m + 1) + kT[0][s]++] = -1;
      ++cnt;
      eigvals[s].push_back(0.0);
      info.nicolaidesLoc++;
    }
    info.estimDimELoc += cnt + cg;
    ksub[s] = cnt + cg;
    if (ksub[s] > 256) {
      release_all();
      return fail("GenEO: more than 256 coarse vectors in one subdomain: set -geneo_cut or lower -geneo_tau");
    }
  }
  std::vector<int64_t> zbase(ns + 1, 0);
  for (int s = 0; s < ns; ++s) zbase[s + 1] = zbase[s] + (int64_t)ksub[s] * (int64_t)subs[s].l2g.size();
  d_Z = (double*)bk::alloc(sizeof(double) * (size_t)std::max<int64_t>(1, zbase[ns]));
  d_zbase = (int64_t*)bk::alloc(sizeof(int64_t) * (ns + 1));
  bk::h2d(d_zbase, zbase.data(), sizeof(int64_t) * (ns + 1));
  int64_t* dzb = (int64_t*)bk::alloc(sizeof(int64_t) * (ns + 1));
  size_t selmax = 1;
  for (auto& v : selT) selmax = std::max(selmax, v.size());
  for (auto& v : selG) selmax = std::max(selmax, v.size());
  int* dsel = (int*)bk::alloc(sizeof(int) * selmax);
  int* dks = (int*)bk::alloc(sizeof(int) * std::max(1, ns));
  // every stage writes its vectors behind those of the stages before it
  std::vector<int64_t> zcur(zbase.begin(), zbase.end());
  auto extract = [&](Stage& st, const std::vector<int>& sel, const std::vector<int>& ks, const double* colscale) {
    if (colscale) {
      double* dgs = dv((size_t)ns * st.m);
      bk::h2d(dgs, colscale, sizeof(double) * (size_t)ns * st.m);
      bk::block_colscale(ch, st.X, st.m, st.m, dgs);
      bk::sync();
      bk::dfree(dgs);
    }
    bk::h2d(dzb, zcur.data(), sizeof(int64_t) * (ns + 1));
    bk::h2d(dsel, sel.data(), sizeof(int) * sel.size());
    bk::h2d(dks, ks.data(), sizeof(int) * ns);
    bk::block_extract(ch, st.X, st.m, st.m + 1, d_D, dsel, dks, dzb, d_Z);
    bk::sync();
    for (int s = 0; s < ns; ++s) zcur[s] += (int64_t)ks[s] * (int64_t)subs[s].l2g.size();
  };
  for (size_t g = 0; g < stT.size(); ++g) extract(stT[g], selT[g], kT[g], nullptr);
  for (size_t g = 0; g < stG.size(); ++g) extract(stG[g], selG[g], kG[g], gscale[g].data());
  bk::dfree(dzb);
  bk::dfree(dsel);
  bk::dfree(dks);
  release_all();
  return 0;
}

// ------------------------------------------------------------------------------------ memory-bounded set-up
// Which consecutive subdomains are eigensolved together ({0, ns}: all at once, the plain path).  GenEO-1 without
// -geneo_chk only: GenEO-2's gamma problem runs through the level-1 hierarchy and needs the other subdomains' flags.
std::vector<int> PC::plan_eig_groups() const {
  const int ns = (int)subs.size();
  const std::vector<int> all = {0, ns};
  if (eig_only || ns < 2 || opt.lvl2 != 1 || opt.check) return all;
  int nmax = 0;
  for (auto& s : subs) nmax = std::max(nmax, (int)s.l2g.size());
  if (nmax <= 192) return all;
  const int m = eig_block_max();
  int64_t cap = opt.eig_group_rows;
  if (cap <= 0) {
    double budget = opt.eig_mem_gb * 1073741824.0;
    if (budget <= 0.0) {
      double total = 0.0;
      bk::mem_info(nullptr, nullptr, nullptr, nullptr, nullptr, &total, false);
      if (total <= 0.0) return all;
      budget = 0.35 * total;
    }
    // six n x 3m basis blocks when the iteration starts, three n x m residual / direction blocks, the V-cycle's block
    // work space (~4 m per fine row over all levels), the Ritz vectors, the hierarchy and the fine matrices themselves
    const double per_row = 8.0 * 30.0 * m + 700.0;
    cap = (int64_t)(budget / per_row);
  }
  cap = std::max<int64_t>(1, std::min<int64_t>(cap, (int64_t)0x7fffffff / (3 * m) - 1));   // n x 3m blocks: 32-bit element indices
  // the fewest groups the cap allows (greedy), then the same number of groups with the rows spread evenly: equal
  // subdomains give equal groups, whose device blocks the caching allocator hands from one group to the next
  auto split = [&](int64_t lim) {
    std::vector<int> gb = {0};
    int64_t rows = 0;
    for (int s = 0; s < ns; ++s) {
      const int64_t n = (int64_t)subs[s].l2g.size();
      if (rows > 0 && rows + n > lim) {
        gb.push_back(s);
        rows = 0;
      }
      rows += n;
    }
    gb.push_back(ns);
    return gb;
  };
  std::vector<int> gb = split(cap);
  const int ng = (int)gb.size() - 1;
  if (ng > 1) {
    int64_t total = 0, nmax64 = 0;
    for (auto& s : subs) { total += (int64_t)s.l2g.size(); nmax64 = std::max<int64_t>(nmax64, (int64_t)s.l2g.size()); }
    for (int64_t lim = std::max(nmax64, (total + ng - 1) / ng); lim <= cap; lim += std::max<int64_t>(1, nmax64 / 8)) {
      std::vector<int> even = split(lim);
      if ((int)even.size() - 1 == ng) { gb = even; break; }
    }
  }
  return gb;
}

// The eigensolve of this rank's subdomains group by group.  Each group is handed to a temporary PC (eig_only) as a
// problem of its own -- its subdomains in local numbering, their matrices MOVED in and back, never copied --, which runs
// the ordinary set-up up to the end of eigen_lobpcg and is destroyed before the next group starts: its fine matrices, its
// A_Neu hierarchy and the LOBPCG blocks are all that ever coexist with this PC's resident matrices.  A subdomain's
// iteration depends on nothing outside its own rows (start block from its global id and local row, per-subdomain
// Rayleigh-Ritz, per-subdomain freezing), so eigenvalues, kept counts and Z are those of the all-at-once path.
int PC::eigen_grouped() {
  const int ns = (int)subs.size();
  const int ng = (int)eig_groups.size() - 1;
  ksub.assign(ns, 0);
  info.eig_iterations = 0;
  info.eig_coarse_iterations = 0;
  std::vector<double*> zpiece(ng, nullptr);
  auto drop_pieces = [&]() {
    for (double* p : zpiece) bk::dfree(p);
  };
  struct GroupRun {
    PC q;
    int s0 = 0, s1 = 0, rc = 1;
    std::string err;
  };
  auto move_subs = [&](GroupRun& r, bool back) {
    for (int s = r.s0; s < r.s1; ++s) {
      Sub &a = subs[s], &b = r.q.subs[s - r.s0];
      std::swap(a.mult, b.mult);
      std::swap(a.a_neu, b.a_neu);
      std::swap(a.a_dir, b.a_dir);
      if (!back) {
        b.gid = a.gid;
        b.l2g.resize(a.l2g.size());
        std::iota(b.l2g.begin(), b.l2g.end(), suboff[s] - suboff[r.s0]);
      }
    }
  };
  auto make = [&](int g) {
    std::unique_ptr<GroupRun> r(new GroupRun());
    r->s0 = eig_groups[g];
    r->s1 = eig_groups[g + 1];
    r->q.opt = opt;
    r->q.eig_only = true;
    r->q.nsub_global = nsub_global;
    r->q.N = suboff[r->s1] - suboff[r->s0];
    r->q.subs.resize(r->s1 - r->s0);
    move_subs(*r, false);
    return r;
  };
  auto guarded = [](GroupRun& r, const std::function<int()>& f) {
    try {
      r.rc = f();
      if (r.rc) r.err = r.q.last_error;
    } catch (std::exception& e) {
      r.rc = 1;
      r.err = e.what();
    }
  };
  // Two groups in flight: while the main stream iterates on group g, a helper thread on a side stream uploads the
  // matrices of group g + 1 and builds its A_Neu hierarchy (setup_prepare: host copies into pinned memory, host
  // aggregation, downloads of the coarse matrices -- 0.15 of the 0.41 s a 6.5 M-row group takes, mostly host-bound).
  // GENEO_EIG_PIPELINE=0: one group at a time.
  static const bool pipelined = !(getenv("GENEO_EIG_PIPELINE") && !strcmp(getenv("GENEO_EIG_PIPELINE"), "0"));
  std::unique_ptr<GroupRun> cur = make(0), nxt;
  guarded(*cur, [&]() { return cur->q.setup_prepare(); });
  for (int g = 0; g < ng; ++g) {
    std::thread helper;
    if (g + 1 < ng) {
      nxt = make(g + 1);
      GroupRun* nr = nxt.get();
      if (pipelined)
        helper = std::thread([nr, &guarded]() {
          bk::side_stream_begin();
          guarded(*nr, [nr]() { return nr->q.setup_prepare(); });
          bk::side_stream_end();
        });
    }
    if (!cur->rc) guarded(*cur, [&]() { return cur->q.setup_finish(nullptr); });
    if (helper.joinable()) helper.join();
    else if (nxt) guarded(*nxt, [&]() { return nxt->q.setup_prepare(); });
    move_subs(*cur, true);
    if (cur->rc) {
      if (nxt) move_subs(*nxt, true);
      drop_pieces();
      return fail(cur->err.empty() ? "GenEO: grouped eigensolve failed" : cur->err);
    }
    PC& q = cur->q;
    const int s0 = cur->s0, s1 = cur->s1;
    for (int s = s0; s < s1; ++s) {
      eigvals[s] = q.eigvals[s - s0];
      candidates[s] = q.candidates[s - s0];
      ksub[s] = q.ksub[s - s0];
    }
    info.nicolaidesLoc += q.info.nicolaidesLoc;
    info.estimDimELoc += q.info.estimDimELoc;
    info.eig_iterations = std::max(info.eig_iterations, q.info.eig_iterations);   // as in one batch: the slowest subdomain's
    info.eig_coarse_iterations = std::max(info.eig_coarse_iterations, q.info.eig_coarse_iterations);
    info.eig_spmm += q.info.eig_spmm;
    info.amgSetupTime += q.info.amgSetupTime;
    if (g == 0) {
      info.amg_on_device = q.info.amg_on_device;
      info.amg_levels = q.info.amg_levels;
      info.amg_operator_complexity = q.info.amg_operator_complexity;
    }
    zpiece[g] = q.d_Z;            // column-major Z_s of the group's subdomains, back to back
    q.d_Z = nullptr;
    if (getenv("GENEO_DEBUG"))
      fprintf(stderr, "[setup] eigensolve group %d of %d: subdomains %d..%d, %d rows, %d LOBPCG iterations, set-up of the group %.3f s (prepared %s)\n",
              g + 1, ng, s0, s1 - 1, q.N, q.info.eig_iterations, q.info.setupTime,
              (pipelined && g > 0) ? "behind the previous group's iteration" : "in line");
    cur = std::move(nxt);
  }
  std::vector<int64_t> zbase(ns + 1, 0);
  for (int s = 0; s < ns; ++s) {
    if (ksub[s] > 256) { drop_pieces(); return fail("GenEO: more than 256 coarse vectors in one subdomain: set -geneo_cut or lower -geneo_tau"); }
    zbase[s + 1] = zbase[s] + (int64_t)ksub[s] * (int64_t)subs[s].l2g.size();
  }
  d_Z = (double*)bk::alloc(sizeof(double) * (size_t)std::max<int64_t>(1, zbase[ns]));
  d_zbase = (int64_t*)bk::alloc(sizeof(int64_t) * (ns + 1));
  bk::h2d(d_zbase, zbase.data(), sizeof(int64_t) * (ns + 1));
  for (int g = 0; g < ng; ++g) {
    const int64_t b0 = zbase[eig_groups[g]], b1 = zbase[eig_groups[g + 1]];
    if (b1 > b0) bk::d2d(d_Z + b0, zpiece[g], sizeof(double) * (size_t)(b1 - b0));
  }
  bk::sync();
  drop_pieces();
  return 0;
}

// ------------------------------------------------------------------------------------ -geneo_chk
// Diagnostics of the reference's --check mode (geneo.cpp:173-247 checkRank, :782-840 checkSPD, :988-997
// partition of unity).  Files are written in the working directory under the reference's names;
// the id in "check<id>" is the global subdomain id (= the MPI rank of the reference).
static std::string check_id(int gid, int nsub) {
  std::ostringstream w, r;
  w << nsub;
  r << std::setfill('0') << std::setw((int)w.str().length()) << gid;
  return "check" + r.str();
}

// checkSPD(false, pcA, "check", "A") (geneo.cpp:1740-1743): smallest Ritz values of a Lanczos run on the
// assembled operator (the reference asks ARPACK for the smallest-magnitude eigenvalue, then an LDLt
// inertia that this path does not have: the inertia line says so).
int PC::check_global_spd() {
  const int n = n_owned();
  const int steps = std::max(1, std::min(N, 80));
  auto dv = [](size_t k) { return (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, k)); };
  double *v = dv(n), *vp = dv(n), *w = dv(n);
  std::vector<double> h(std::max(1, n));
  for (int i = 0; i < n; ++i) h[i] = bk::hash_unit_host(opt.eps_seed + 17, 0, (uint64_t)owned[i], 0) - 0.5;
  bk::h2d(v, h.data(), sizeof(double) * n);
  bk::set(vp, 0.0, n);
  double nrm = std::sqrt(gdot(v, v));
  bk::axpby(v, 0.0, v, 1.0 / nrm, n);
  std::vector<double> al, be;
  double beta = 0.0;
  int rc = 0;
  for (int k = 0; k < steps; ++k) {
    if ((rc = matmult(v, w))) break;
    const double a = gdot(w, v);
    al.push_back(a);
    bk::axpy(w, -a, v, n);
    bk::axpy(w, -beta, vp, n);
    beta = std::sqrt(gdot(w, w));
    if (!(beta > 1e-13 * std::fabs(a))) break;   // invariant subspace: T is exact
    be.push_back(beta);
    bk::copy(vp, v, n);
    bk::copy(v, w, n);
    bk::axpby(v, 0.0, v, 1.0 / beta, n);
  }
  bk::dfree(v); bk::dfree(vp); bk::dfree(w);
  if (rc) return rc;
  const int k = (int)al.size();
  std::vector<double> T((size_t)k * k, 0.0), ev, V;
  for (int i = 0; i < k; ++i) {
    T[(size_t)i * k + i] = al[i];
    if (i + 1 < k && i < (int)be.size()) T[(size_t)i * k + i + 1] = T[(size_t)(i + 1) * k + i] = be[i];
  }
  dense::sym_eig(T, k, ev, V);
  const double lmin = *std::min_element(ev.begin(), ev.end());
  if (rank == 0) {
    std::ofstream f("check.SPD.A.log");
    f << "A - eigen value 0: " << lmin << std::endl;
    f << std::endl << "A - inertia: not computed (no LDLt on the MI355X path; " << k << " Lanczos steps, largest Ritz value "
      << *std::max_element(ev.begin(), ev.end()) << ")" << std::endl;
  }
  if (lmin <= DBL_EPSILON) {
    std::ostringstream msg;
    msg << "GenEO - check SPD: A not SPD, bad eigen value " << lmin;
    return fail(msg.str());
  }
  return 0;
}

// checkSPD(true, pcBLoc, checkFile, pb + ".B") (geneo.cpp:883-886): smallest eigenvalue of the right-hand
// matrix of each local pencil by LOBPCG on (B, I); exact inertia from a dense eigen-decomposition while
// the subdomain is small enough.
int PC::check_local_spd(const EigProblem& P, const HostCsr* const* hostB, bool scale_mult) {
  const int ns = (int)subs.size();
  // identity as the second operator of the pencil
  std::vector<int> rp(nL + 1), col(std::max(1, nL));
  std::vector<double> val(std::max(1, nL), 1.0);
  for (int i = 0; i <= nL; ++i) rp[i] = i;
  for (int i = 0; i < nL; ++i) col[i] = i;
  bk::Csr eye = bk::csr_upload(nL, rp.data(), col.data(), val.data());
  EigProblem Q{P.B, P.Bs, &eye, nullptr, nullptr, nullptr, 0.0, 1, P.label};
  // Jacobi scaling and Gershgorin bound of B for the Chebyshev preconditioner
  std::vector<double> dinv(std::max(1, nL));
  double lmax = 0.0;
  for (int s = 0; s < ns; ++s) {
    const HostCsr& b = *hostB[s];
    const std::vector<int>& mu = subs[s].mult;
    for (int i = 0; i < b.n; ++i) {
      double diag = 0.0, row = 0.0;
      for (int k = b.rowptr[i]; k < b.rowptr[i + 1]; ++k) {
        const double v = scale_mult ? b.val[k] / ((double)mu[i] * (double)mu[b.col[k]]) : b.val[k];
        if (b.col[k] == i) diag += v;
        row += std::fabs(v);
      }
      if (!(diag > 0.0)) {
        bk::csr_free(eye);
        std::ostringstream msg;
        msg << "GenEO - check SPD: " << P.label << ".B not SPD, bad diagonal value " << diag;
        return fail(msg.str());
      }
      dinv[suboff[s] + i] = 1.0 / diag;
      lmax = std::max(lmax, row / diag);
    }
  }
  double* d_dinv = (double*)bk::alloc(sizeof(double) * std::max(1, nL));
  bk::h2d(d_dinv, dinv.data(), sizeof(double) * nL);
  Q.dinv = d_dinv;
  Q.lmax = lmax;
  std::vector<double> lam;
  double* X = (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, (size_t)nL * 16));
  const int its_before = info.eig_iterations;
  int rc = lobpcg_solve(Q, 16, lam, X);
  info.eig_iterations = its_before;   // diagnostics do not count as eigensolver work
  bk::dfree(X);
  bk::dfree(d_dinv);
  bk::csr_free(eye);
  if (rc) return rc;
  for (int s = 0; s < ns; ++s) {
    const std::string name = check_id(subs[s].gid, nsub_global) + ".SPD." + P.label + ".B.log";
    std::ofstream f(name.c_str());
    const double l0 = lam[(size_t)s * 16];
    f << P.label << ".B - eigen value 0: " << l0 << std::endl;
    const int n = (int)subs[s].l2g.size();
    int neg = 0, nul = 0, pos = 0;
    bool have_inertia = false;
    if (n <= 1024) {
      const HostCsr& b = *hostB[s];
      std::vector<double> G((size_t)n * n, 0.0), w, V;
      for (int i = 0; i < n; ++i)
        for (int k = b.rowptr[i]; k < b.rowptr[i + 1]; ++k)
          G[(size_t)i * n + b.col[k]] += scale_mult ? b.val[k] / ((double)subs[s].mult[i] * (double)subs[s].mult[b.col[k]]) : b.val[k];
      dense::sym_eig(G, n, w, V);
      double wmax = 0.0;
      for (double v : w) wmax = std::max(wmax, std::fabs(v));
      for (double v : w) {
        if (std::fabs(v) <= 1e-14 * wmax) nul++;
        else if (v < 0) neg++;
        else pos++;
      }
      have_inertia = true;
      f << std::endl << P.label << ".B - inertia: nbNegEV " << neg << ", nbNullEV " << nul << ", nbPosEV " << pos << std::endl;
    } else {
      f << std::endl << P.label << ".B - inertia: not computed (n = " << n << " > 1024, no LDLt on the MI355X path)" << std::endl;
    }
    if (l0 <= DBL_EPSILON) {
      std::ostringstream msg;
      msg << "GenEO - check SPD: " << P.label << ".B not SPD, bad eigen value " << l0;
      return fail(msg.str());
    }
    if (have_inertia && (neg > 0 || nul > 0))
      return fail("GenEO - check SPD: not SPD (inertia - negative or null eigen value found)");
  }
  return 0;
}

// checkRank(true, pcZE2L, ...) (geneo.cpp:280-283): Z_s = Q R by modified Gram-Schmidt with one
// refinement pass (BV_ORTHOG_MGS + REFINE_ALWAYS, :203); every R(i,i) must be non zero.
int PC::check_local_rank() {
  const int ns = (int)subs.size();
  std::vector<int64_t> zb(ns + 1);
  bk::d2h(zb.data(), d_zbase, sizeof(int64_t) * (ns + 1));
  for (int s = 0; s < ns; ++s) {
    const int n = (int)subs[s].l2g.size(), k = ksub[s];
    std::vector<double> Zs((size_t)n * k), R((size_t)k * k, 0.0);
    bk::d2h(Zs.data(), d_Z + zb[s], sizeof(double) * (size_t)n * k);
    for (int j = 0; j < k; ++j) {
      double* zj = Zs.data() + (size_t)j * n;
      for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < j; ++i) {
          const double* qi = Zs.data() + (size_t)i * n;
          double d = 0.0;
          for (int t = 0; t < n; ++t) d += qi[t] * zj[t];
          for (int t = 0; t < n; ++t) zj[t] -= d * qi[t];
          R[(size_t)i * k + j] += d;
        }
      double nr = 0.0;
      for (int t = 0; t < n; ++t) nr += zj[t] * zj[t];
      nr = std::sqrt(nr);
      R[(size_t)j * k + j] = nr;
      if (nr > 0)
        for (int t = 0; t < n; ++t) zj[t] /= nr;
    }
    const std::string name = check_id(subs[s].gid, nsub_global) + ".setup.Z.R";
    std::ofstream f(name.c_str());
    f.precision(16);
    for (int i = 0; i < k; ++i) {
      for (int j = 0; j < k; ++j) f << (j ? " " : "") << R[(size_t)i * k + j];
      f << std::endl;
    }
    for (int i = 0; i < k; ++i)
      if (std::fabs(R[(size_t)i * k + i]) <= DBL_EPSILON) {
        std::ostringstream msg;
        msg << "GenEO - check rank: Z = Q*R with R(" << i << ", " << i << ") = " << R[(size_t)i * k + i];
        return fail(msg.str());
      }
  }
  return 0;
}

// checkRank(false, pcZE2G, ...) (geneo.cpp:412-415): the global Z (N x dimE).  R is the Cholesky factor of
// the Gram matrix Z^T Z assembled with the operators of the coarse correction (Z, R^T, R, Z^T).
int PC::check_global_rank() {
  std::vector<double> G((size_t)dimE * dimE, 0.0), unit(dimE, 0.0), colv(dimE);
  for (int j = 0; j < dimE; ++j) {
    unit[j] = 1.0;
    bk::h2d(d_yE, unit.data(), sizeof(double) * dimE);
    unit[j] = 0.0;
    bk::z_apply(ch, d_Z, d_zbase, d_ksub, d_zoff, d_yE, d_wL, false);
    prolong_add(d_wL, d_t1);
    restrict_to_local(d_t1, d_xL);
    bk::zt_apply(ch, d_Z, d_zbase, d_ksub, d_zoff, kmax, d_xL, d_yE, dimE);
    allreduce(d_yE, dimE);
    bk::d2h(colv.data(), d_yE, sizeof(double) * dimE);
    for (int i = 0; i < dimE; ++i) G[(size_t)i * dimE + j] = colv[i];
  }
  // right-looking Cholesky G = R^T R; a non-positive pivot is reported as R(i,i) = 0
  std::vector<double> R((size_t)dimE * dimE, 0.0);
  int bad = -1;
  for (int i = 0; i < dimE && bad < 0; ++i) {
    double d = G[(size_t)i * dimE + i];
    for (int t = 0; t < i; ++t) d -= R[(size_t)t * dimE + i] * R[(size_t)t * dimE + i];
    if (!(d > DBL_EPSILON * DBL_EPSILON * G[(size_t)i * dimE + i])) { bad = i; break; }
    const double rii = std::sqrt(d);
    R[(size_t)i * dimE + i] = rii;
    for (int j = i + 1; j < dimE; ++j) {
      double v = 0.5 * (G[(size_t)i * dimE + j] + G[(size_t)j * dimE + i]);
      for (int t = 0; t < i; ++t) v -= R[(size_t)t * dimE + i] * R[(size_t)t * dimE + j];
      R[(size_t)i * dimE + j] = v / rii;
    }
  }
  if (rank == 0) {
    std::ofstream f("check.setup.ZE2G.R");
    f.precision(16);
    for (int i = 0; i < dimE; ++i) {
      for (int j = 0; j < dimE; ++j) f << (j ? " " : "") << R[(size_t)i * dimE + j];
      f << std::endl;
    }
  }
  if (bad >= 0) {
    std::ostringstream msg;
    msg << "GenEO - check rank: Z = Q*R with R(" << bad << ", " << bad << ") = 0";
    return fail(msg.str());
  }
  return 0;
}

// E = Z^T A Z (createEEig, geneo.cpp:1028-1095), replicated on every rank, dense factorisation on host.
int PC::build_E() {
  const int ns = (int)subs.size();
  // all_gather of the local sizes (createZE2G, geneo.cpp:360-375) through one all-reduce
  std::vector<double> kg(std::max(1, nsub_global), 0.0);
  for (int s = 0; s < ns; ++s) {
    if (subs[s].gid < 0 || subs[s].gid >= nsub_global) return fail("GenEO: bad global subdomain id");
    kg[subs[s].gid] = ksub[s];
  }
  if (size > 1) {
    double* dk = (double*)bk::alloc(sizeof(double) * nsub_global);
    bk::h2d(dk, kg.data(), sizeof(double) * nsub_global);
    allreduce(dk, nsub_global);
    bk::d2h(kg.data(), dk, sizeof(double) * nsub_global);
    bk::dfree(dk);
  }
  ksub_global.assign(nsub_global, 0);
  std::vector<int> zoff_g(nsub_global + 1, 0);
  for (int g = 0; g < nsub_global; ++g) {
    ksub_global[g] = (int)std::lround(kg[g]);
    zoff_g[g + 1] = zoff_g[g] + ksub_global[g];
  }
  dimE = zoff_g[nsub_global];
  info.dimE = dimE;
  info.realDimELoc = 0;
  kmax = 0;
  zoff.assign(ns, 0);
  for (int s = 0; s < ns; ++s) {
    zoff[s] = zoff_g[subs[s].gid];
    kmax = std::max(kmax, ksub[s]);
    info.realDimELoc += ksub[s];
  }
  d_ksub = (int*)bk::alloc(sizeof(int) * std::max(1, ns));
  d_zoff = (int*)bk::alloc(sizeof(int) * std::max(1, ns));
  bk::h2d(d_ksub, ksub.data(), sizeof(int) * ns);
  bk::h2d(d_zoff, zoff.data(), sizeof(int) * ns);
  d_yE = (double*)bk::alloc(sizeof(double) * std::max(1, dimE));
  h_yE.assign(std::max(1, dimE), 0.0);
  E.assign((size_t)dimE * dimE, 0.0);
  const int nown = n_owned();
  const int W = std::min(32, size == 1 ? 32 : comm_width);
  if (W >= 8 && dimE > 0 && !getenv("GENEO_E_COLUMNWISE")) {
    // Blocked assembly: W columns of E per pass, 4 wide halo exchanges per pass instead of 4 per column.
    //   WL = ZR C_j (selection of W coarse vectors, MFMA block kernel) ; T1 = sum R^T WL ; T2 = A T1 ;
    //   XL = R T2 ; G_s = ZR_s^T XL_s (MFMA Gram) = the rows of E owned by subdomain s, columns of the pass
    const int kp = ((std::max(1, kmax) + 15) / 16) * 16;
    auto dv = [](size_t n) { return (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, n)); };
    double* ZR = dv((size_t)nL * kp);
    double *WLb = dv((size_t)nL * W), *XLb = dv((size_t)nL * W);
    double *T1 = dv((size_t)nown * W), *T2 = dv((size_t)nown * W);
    double* xe = dv((size_t)std::max(nE, nown) * W);
    double *dC = dv((size_t)ns * kp * W), *dG = dv((size_t)ns * kp * W);
    std::vector<double> hC((size_t)ns * kp * W), hG((size_t)ns * kp * W);
    bk::z_rowmajor(ch, d_Z, d_zbase, d_ksub, ZR, kp);
    int rc = 0;
    for (int j0 = 0; j0 < dimE && !rc; j0 += W) {
      std::fill(hC.begin(), hC.end(), 0.0);
      for (int s = 0; s < ns; ++s)
        for (int k = 0; k < ksub[s]; ++k) {
          const int c = zoff[s] + k - j0;
          if (c >= 0 && c < W) hC[((size_t)s * kp + k) * W + c] = 1.0;
        }
      bk::h2d(dC, hC.data(), sizeof(double) * hC.size());
      bk::block_mul(ch, ZR, kp, kp, dC, W, WLb, W, false);
      prolong_block(WLb, T1, W, xe);
      matmult_block(T1, T2, W, WLb, xe);
      restrict_block(T2, XLb, W, xe);
      bk::gram(ch, ZR, kp, kp, XLb, W, W, dG);
      bk::d2h(hG.data(), dG, sizeof(double) * hG.size());
      for (int s = 0; s < ns; ++s)
        for (int k = 0; k < ksub[s]; ++k)
          for (int c = 0; c < W && j0 + c < dimE; ++c)
            E[(size_t)(zoff[s] + k) * dimE + j0 + c] = hG[((size_t)s * kp + k) * W + c];
    }
    for (double* p : {ZR, WLb, XLb, T1, T2, xe, dC, dG}) bk::dfree(p);
    if (size > 1) {  // rows of E live with the rank of their subdomain: one sum over the ranks
      double* dE = dv(E.size());
      bk::h2d(dE, E.data(), sizeof(double) * E.size());
      allreduce(dE, (int)E.size());
      bk::d2h(E.data(), dE, sizeof(double) * E.size());
      bk::dfree(dE);
    }
  } else {
  // column j of E: Z^T A (Z e_j)
  std::vector<double> unit(dimE, 0.0), colv(dimE);
  for (int j = 0; j < dimE; ++j) {
    unit[j] = 1.0;
    bk::h2d(d_yE, unit.data(), sizeof(double) * dimE);
    unit[j] = 0.0;
    bk::z_apply(ch, d_Z, d_zbase, d_ksub, d_zoff, d_yE, d_wL, false);
    prolong_add(d_wL, d_t1);
    if (int rc = matmult(d_t1, d_t2)) return rc;
    restrict_to_local(d_t2, d_xL);
    bk::zt_apply(ch, d_Z, d_zbase, d_ksub, d_zoff, kmax, d_xL, d_yE, dimE);
    allreduce(d_yE, dimE);
    bk::d2h(colv.data(), d_yE, sizeof(double) * dimE);
    for (int i = 0; i < dimE; ++i) E[(size_t)i * dimE + j] = colv[i];
  }
  }
  (void)nown;
  Efac = E;
  for (int a = 0; a < dimE; ++a)
    for (int b = a + 1; b < dimE; ++b) Efac[(size_t)a * dimE + b] = Efac[(size_t)b * dimE + a] =
        0.5 * (Efac[(size_t)a * dimE + b] + Efac[(size_t)b * dimE + a]);
  std::vector<double> sym = Efac;
  E_chol = dense::cholesky_blocked(Efac, dimE, (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
  if (E_chol) {   // U = L^T next to L: the backward substitution then walks contiguous rows too
    EfacT.assign(Efac.size(), 0.0);
    for (int a = 0; a < dimE; ++a)
      for (int b = 0; b <= a; ++b) EfacT[(size_t)b * dimE + a] = Efac[(size_t)a * dimE + b];
  }
  if (E_chol && dimE > 0 && dimE <= 1024) {     // the factor, twice, for the device sweeps of coarse_solve_local
    d_EL = (double*)bk::alloc(sizeof(double) * Efac.size());
    d_ELT = (double*)bk::alloc(sizeof(double) * EfacT.size());
    bk::h2d(d_EL, Efac.data(), sizeof(double) * Efac.size());
    bk::h2d(d_ELT, EfacT.data(), sizeof(double) * EfacT.size());
  }
  if (!E_chol) {
    Efac = sym;
    if (!dense::lu_factor(Efac, dimE, Epiv)) return fail("GenEO - solve KO: dcs2 (singular coarse operator E)");
  }
  return 0;
}

// ------------------------------------------------------------------------------------ Krylov
double PC::gdot(const double* x, const double* y) {
  bk::dot(x, y, n_owned(), d_scal);
  double v = 0.0;
  if (size > 1) {
    allreduce(d_scal, 1);
  }
  bk::d2h(&v, d_scal, sizeof(double));
  return v;
}

// KSPConvergedDefault (PETSc iterativ.c), preconditioned norm
struct ConvTest {
  double rtol, atol, dtol, rnorm0 = 0, ttol = 0;
  int operator()(int it, double rnorm, double snorm_if_guess) {
    if (it == 0) {
      rnorm0 = (snorm_if_guess >= 0.0) ? (snorm_if_guess == 0.0 ? rnorm : snorm_if_guess) : rnorm;
      ttol = std::max(rtol * rnorm0, atol);
    }
    if (std::isnan(rnorm) || std::isinf(rnorm)) return -9;
    if (rnorm <= ttol) return rnorm < atol ? 3 : 2;
    if (rnorm >= dtol * rnorm0) return -4;
    return 0;
  }
};

// Device vectors of a Krylov driver: released on every way out, a throwing callback (halo exchange, all-reduce) included
struct DeviceVectors {
  std::vector<double*> v;
  double* get(int n) {
    v.push_back((double*)bk::alloc(sizeof(double) * std::max(1, n)));
    return v.back();
  }
  ~DeviceVectors() {
    for (double* p : v) bk::dfree(p);
  }
};

// KSPSolve_CG (PETSc cg.c), KSP_NORM_PRECONDITIONED
int PC::solve_cg(const double* b, double* x, KspResult* res) {
  const int n = n_owned();
  DeviceVectors bufs;
  auto dv = [&](void) { return bufs.get(n); };
  double *r = dv(), *z = dv(), *p = dv(), *w = dv();
  auto done = [&](int rc) { return rc; };
  ConvTest conv{opt.ksp_rtol, opt.ksp_atol, opt.ksp_dtol};
  residual_history.clear();
  if (opt.ksp_guess_nonzero) {
    if (int rc = matmult(x, r)) return done(rc);
    bk::axpby(r, 1.0, b, -1.0, n);  // r = b - A x
  } else {
    bk::copy(r, b, n);
  }
  if (int rc = apply(r, z)) return done(rc);
  double dp = std::sqrt(gdot(z, z));
  double snorm = -1.0;
  if (opt.ksp_guess_nonzero) {
    if (int rc = apply(b, w)) return done(rc);
    snorm = std::sqrt(gdot(w, w));
  }
  residual_history.push_back(dp);
  res->its = 0;
  res->rnorm = dp;
  res->reason = conv(0, dp, snorm);
  if (res->reason) return done(0);
  double betaold = 0.0;
  for (int i = 0; i < opt.ksp_max_it; ++i) {
    res->its = i + 1;
    const double beta = gdot(z, r);
    if (beta == 0.0) { res->reason = 3; return done(0); }
    if (i == 0) bk::copy(p, z, n);
    else bk::axpby(p, 1.0, z, beta / betaold, n);  // p = z + b p
    if (int rc = matmult(p, w)) return done(rc);
    const double dpi = gdot(p, w);
    betaold = beta;
    if (!(dpi > 0.0)) { res->reason = -8; return done(0); }  // KSP_DIVERGED_INDEFINITE_MAT
    const double a = beta / dpi;
    bk::axpy(x, a, p, n);
    bk::axpy(r, -a, w, n);
    if (int rc = apply(r, z)) return done(rc);
    dp = std::sqrt(gdot(z, z));
    residual_history.push_back(dp);
    res->rnorm = dp;
    res->reason = conv(i + 1, dp, -1.0);
    if (res->reason) return done(0);
  }
  res->reason = -3;
  return done(0);
}

// KSPSolve_GMRES (PETSc gmres.c): left preconditioning, classical Gram-Schmidt, Givens residual
int PC::solve_gmres(const double* b, double* x, KspResult* res) {
  const int n = n_owned();
  const int m = std::max(1, opt.ksp_restart);
  std::vector<double*> V;
  DeviceVectors bufs;
  auto dvec = [&]() { return bufs.get(n); };
  double *t = dvec(), *w = dvec();
  auto done = [&](int rc) { return rc; };
  ConvTest conv{opt.ksp_rtol, opt.ksp_atol, opt.ksp_dtol};
  residual_history.clear();
  res->its = 0;
  bool first = true;
  std::vector<double> h((size_t)(m + 1) * m), cs(m), sn(m), g(m + 1), yk(m);
  while (true) {
    if (V.empty()) V.push_back(dvec());
    // r = M^-1 (b - A x)
    if (opt.ksp_guess_nonzero || !first) {
      if (int rc = matmult(x, t)) return done(rc);
      bk::axpby(t, 1.0, b, -1.0, n);
      if (int rc = apply(t, V[0])) return done(rc);
    } else {
      if (int rc = apply(b, V[0])) return done(rc);
    }
    double rn = std::sqrt(gdot(V[0], V[0]));
    if (first) {
      double snorm = -1.0;
      if (opt.ksp_guess_nonzero) {
        if (int rc = apply(b, w)) return done(rc);
        snorm = std::sqrt(gdot(w, w));
      }
      residual_history.push_back(rn);
      res->rnorm = rn;
      res->reason = conv(0, rn, snorm);
      if (res->reason) return done(0);
      first = false;
    }
    if (rn == 0.0) { res->reason = 3; return done(0); }
    bk::axpby(V[0], 1.0 / rn, V[0], 0.0, n);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = rn;
    int k = 0;
    int reason = 0;
    while (k < m && res->its < opt.ksp_max_it) {
      if ((int)V.size() < k + 2) V.push_back(dvec());
      if (int rc = matmult(V[k], t)) return done(rc);
      if (int rc = apply(t, w)) return done(rc);
      for (int j = 0; j <= k; ++j) h[(size_t)j * m + k] = gdot(V[j], w);  // classical GS: all dots first
      for (int j = 0; j <= k; ++j) bk::axpy(w, -h[(size_t)j * m + k], V[j], n);
      const double hn = std::sqrt(gdot(w, w));
      h[(size_t)(k + 1) * m + k] = hn;
      if (hn != 0.0) bk::axpby(V[k + 1], 1.0 / hn, w, 0.0, n);
      for (int j = 0; j < k; ++j) {
        const double a = h[(size_t)j * m + k], c = h[(size_t)(j + 1) * m + k];
        h[(size_t)j * m + k] = cs[j] * a + sn[j] * c;
        h[(size_t)(j + 1) * m + k] = -sn[j] * a + cs[j] * c;
      }
      const double den = std::hypot(h[(size_t)k * m + k], h[(size_t)(k + 1) * m + k]);
      cs[k] = h[(size_t)k * m + k] / den;
      sn[k] = h[(size_t)(k + 1) * m + k] / den;
      h[(size_t)k * m + k] = den;
      h[(size_t)(k + 1) * m + k] = 0.0;
      g[k + 1] = -sn[k] * g[k];
      g[k] = cs[k] * g[k];
      rn = std::fabs(g[k + 1]);
      ++k;
      res->its++;
      residual_history.push_back(rn);
      res->rnorm = rn;
      reason = conv(res->its, rn, -1.0);
      if (reason) break;
    }
    for (int i = k - 1; i >= 0; --i) {
      double s = g[i];
      for (int j = i + 1; j < k; ++j) s -= h[(size_t)i * m + j] * yk[j];
      yk[i] = s / h[(size_t)i * m + i];
    }
    for (int j = 0; j < k; ++j) bk::axpy(x, yk[j], V[j], n);
    if (reason) { res->reason = reason; return done(0); }
    if (res->its >= opt.ksp_max_it) { res->reason = -3; return done(0); }
  }
}

int PC::solve(const double* b, double* x, KspResult* res) {
  if (!is_setup) return fail("GenEO preconditioner is not set up");
  auto t0 = clk::now();
  int rc = 0;
  try {
    rc = (opt.ksp_type == "cg") ? solve_cg(b, x, res) : solve_gmres(b, x, res);
  } catch (std::exception& e) {
    return fail(e.what());
  }
  bk::sync();
  info.solveTime = secs(t0, clk::now());
  return rc;
}

}  // namespace geneo
