// GenEO preconditioner core (host orchestration over the device primitives of backend.h).
// MI355X-native counterpart of /root/reference/src/geneo.cpp; see DESIGN.md for the map.
#pragma once
#include <cstdint>
#include <future>
#include <memory>
#include <string>
#include <map>
#include <vector>

#include "backend.h"

namespace geneo {
class AmgDevice;

struct HostCsr {
  int n = 0;
  std::vector<int> rowptr, col;
  std::vector<double> val;
  bool empty() const { return rowptr.empty(); }
};

// -geneo_* options: names, defaults and validation follow geneo.cpp:2329-2514 / :2649-2662.
struct Options {
  bool lvl1ASM = true, lvl1RAS = false, lvl1SRAS = false, lvl1ORAS = false;
  int lvl2 = 1;
  bool hybrid = false, effHybrid = false;
  double optim = 0.0, tau = 0.1, gamma = 10.0;
  bool cst = false;
  int cut = -1;
  bool noSyl = false, offload = false;
  bool check = false;      // -geneo_chk (geneo.cpp:2466-2479)
  // -geneo_nicolaides_zero <x>: the Nicolaides rule (geneo.cpp:896-944) takes min(lambda) >= x DBL_EPSILON as "zero has
  // not been found".  1 = the reference's literal test; default 100 (a computed zero eigenvalue of an exactly singular
  // Neumann matrix is a rounding error of either sign around 1e-16 .. 2e-15: core.cpp)
  double nicolaides_zero = 100.0;
  // -els2_ : local eigensolver (LOBPCG on the GPU replaces ARPACK shift-invert, geneo.cpp:626-744)
  double eps_tol = 1e-3;   // EPSSetTolerances default at geneo.cpp:658
  int eps_nev = 16;        // block target when -geneo_cut is not given (no inertia count on the GPU)
  int eps_max_it = 500;
  int eps_block = 0;       // LOBPCG block size m (0 = auto: multiple of 16 >= nev + guard)
  std::string eps_conv = "sinvert";   // convergence test: ARPACK's shift-invert Ritz estimate | plain "residual" (core.cpp)
  int cheb_degree = 6;     // Chebyshev-Jacobi preconditioner inside LOBPCG
  double cheb_ratio = 20.0;
  double rr_drop = 1e-6;   // pivot threshold of the rank-revealing Rayleigh-Ritz (basis conditioning <= 1/drop)
  uint64_t eps_seed = 0;
  // Memory-bounded set-up (-geneo_eig_group_rows / -geneo_eig_mem_gb): LOBPCG carries ~30 m doubles per local row (m = block
  // width; 7.7 KB per row at m = 32) next to the A_Neu hierarchy -- eight 6.5 M-row subdomains of the 368^3 benchmark on
  // ONE GPU would need 400 GB.  When the rank's subdomains exceed the budget they are eigensolved in consecutive groups
  // (each group: its own fine matrices, A_Neu hierarchy and basis blocks, released before the next one starts); the
  // per-subdomain iteration is independent of its neighbours in the batch, so the eigenpairs are the ungrouped ones.
  //   eig_group_rows > 0: at most this many local rows per group (a subdomain is never split);  0: from the budget
  //   eig_mem_gb     > 0: device-memory budget of one group's eigensolve in GiB;  0: 35 % of the card (unknown card: one group)
  int eig_group_rows = 0;
  double eig_mem_gb = 0.0;
  // Coarse start of the local eigensolves (-geneo_eig_coarse_start R): when every subdomain of the batch (the rank's
  // subdomains, or one group of the memory-bounded set-up) holds at least R rows, LOBPCG first runs on the Galerkin pencil of multigrid level 1 (A_c = P^T A_Neu P from the hierarchy,
  // B_c = P^T D A_Dir D P: two sparse products) and the fine iteration starts from the prolonged Ritz vectors instead of a
  // random block: an iteration there costs an eighth of a fine one and the fine solve is left with the last digits.  The
  // fine pairs meet the same convergence test either way.  0: never.
  int eig_coarse_start = 750000;
  // -dls1_ : local "direct" solve replaced by batched Jacobi-PCG driven to a tight tolerance
  double dls1_rtol = 1e-12;
  int dls1_max_it = 20000;
  int dls1_check = 16;     // host convergence poll period (iterations; 4 with the AMG preconditioner)
  // inner preconditioners: "amg" = smoothed-aggregation V-cycle (default), "jacobi" / "cheb" = round-1 baseline
  std::string dls1_pc = "amg", els2_pc = "amg";
  int amg_coarse_size = 600, amg_smooth_degree = 1, amg_max_levels = 10;
  double amg_smooth_ratio = 4.0;
  // the same per hierarchy (0: the common -amg_smooth_ratio): -dls1_amg_smooth_ratio for the level-1 hierarchy of the local
  // solves, -els2_amg_smooth_ratio for LOBPCG's A_Neu hierarchy
  // Local solves: 12 -- the damped-Jacobi weight 1 / (0.55 rho (1 + 1 / ratio)) grows from 1.45 / rho (ratio 4) to 1.68 / rho
  // against a Gershgorin bound that overestimates rho(D^-1 A): one rank of 368^3, 496 -> 440 inner iterations, solve 0.391 ->
  // 0.349 s, flat from 12 up (profiles/r04_inner_solver_sweep_rank_of_8.log).  The inner PCG runs to -dls1_ksp_rtol whatever
  // its preconditioner, so the outer iteration is unchanged.  LOBPCG's hierarchy keeps 4: its V-cycle is also the ruler of
  // the convergence test at -els2_eps_tol (8: 19 -> 18 iterations, and a different ruler).
  double dls1_amg_smooth_ratio = 12.0, els2_amg_smooth_ratio = 0.0;
  // -dls1_amg_strength / -els2_amg_strength (AmgParams::strength).  Local solves: 0.04 (126^3: 26 -> 18 inner iterations per
  // solve, same outer counts).  LOBPCG's hierarchy keeps every connection: its V-cycle is also the ruler of the convergence
  // test at -els2_eps_tol, and a different ruler moves the outer iteration count at the loose tolerance of the benchmark
  // (126^3: 25 -> 23) away from the reference-literal oracle's.
  double dls1_amg_strength = 0.04, els2_amg_strength = 0.0;
  bool dls1_amg_single = true;   // -dls1_amg_precision single|double: storage of the level matrices the V-cycle of the local solves reads
  // Krylov driver (counterpart of the PETSc KSP the reference calls at driver:1240)
  std::string ksp_type = "gmres";
  double ksp_rtol = 1e-5, ksp_atol = 1e-50, ksp_dtol = 1e5;
  int ksp_max_it = 10000, ksp_restart = 30;
  bool ksp_guess_nonzero = true;   // driver:1348 always sets it
  std::string name() const;        // buildGenEOName, geneo.cpp:2245-2268
};
// returns "" on success, otherwise the error message (same texts as the reference)
std::string parse_option(Options& o, const std::string& key, const std::string& value);
std::string validate_options(const Options& o);

struct Info {                 // public counters / timers of geneoContext (hdr/geneo.hpp:96-123)
  int estimDimELoc = 0, realDimELoc = 0, nicolaidesLoc = 0, dimE = 0;
  int eig_iterations = 0, eig_spmm = 0;
  long long dls1_iterations = 0, dls1_solves = 0;
  double lvl1SetupMinvTimeLoc = 0, lvl2SetupEigTimeLoc = 0, lvl2SetupZTimeLoc = 0, lvl2SetupETimeLoc = 0;
  double lvl1ApplyTimeLoc = 0, lvl1ApplyScatterTimeLoc = 0, lvl1ApplyMinvTimeLoc = 0, lvl1ApplyGatherTimeLoc = 0;
  double lvl1ApplyPrjFSTimeLoc = 0, lvl2ApplyTimeLoc = 0, lvl2ApplyZtTimeLoc = 0, lvl2ApplyEinvTimeLoc = 0,
         lvl2ApplyZTimeLoc = 0;
  double setupTime = 0, solveTime = 0;
  long long spmv_calls = 0;
  int amg_levels = 0, amg_on_device = 0;
  int eig_groups = 1;          // consecutive subdomain groups the eigensolve ran in (memory-bounded set-up)
  int eig_coarse_iterations = 0;   // LOBPCG iterations on the multigrid level-1 pencil (coarse start; 0: not used)
  double amg_operator_complexity = 0.0, amgSetupTime = 0.0;
  int nullPivotsLoc = 0;
};

struct KspResult {
  int its = 0;
  double rnorm = 0.0;
  int reason = 0;     // PETSc KSPConvergedReason numbering (2 RTOL, 3 ATOL, -3 ITS, -4 DTOL, -9 NANORINF ...)
};

typedef int (*exchange_fn)(void* user, int reverse);
typedef int (*allreduce_fn)(void* user, int n);

class PC {
 public:
  Options opt;
  Info info;
  std::string last_error;

  // ---- inputs (initGenEOPC / PCGenEOSetup, geneo.cpp:2518-2632) ----------------------------
  int N = 0;                 // nbDOF
  int nsub_global = 0;
  int rank = 0, size = 1;
  std::vector<int> owned;    // ascending global ids owned by this rank (all of 0..N-1 when size==1)
  struct Sub {
    int gid = 0;             // global subdomain id (= MPI rank in the reference)
    std::vector<int> l2g;    // dofIdxDomLoc (ascending global ids)
    std::vector<int> mult;   // dofIdxMultLoc
    HostCsr a_neu, a_dir;    // MATIS local matrix / optional pcADirLoc
    std::vector<char> intersect;  // intersectLoc emptiness per global subdomain (GenEO-2 gamma_loc only)
  };
  std::vector<Sub> subs;
  // halo plan (size > 1)
  std::vector<int> halo_gid, recv_counts, send_counts, send_idx;
  exchange_fn cb_exchange = nullptr;
  allreduce_fn cb_allreduce = nullptr;
  void* cb_user = nullptr;
  double *comm_send = nullptr, *comm_recv = nullptr, *comm_red = nullptr;  // device buffers owned by the caller
  int comm_red_cap = 0;
  int comm_width = 1;        // the halo buffers hold this many vectors per exchange (PCGenEOSetCommWidth)

  PC();
  ~PC();
  int add_subdomain(int gid, int n, const int* l2g, const int* mult, const int* neu_rowptr, const int* neu_col,
                    const double* neu_val, const int* dir_rowptr, const int* dir_col, const double* dir_val);
  int set_intersect(int gid, int nb, const int* nonempty);   // intersectLoc of initGenEOPC (hdr/geneo.hpp:34)
  int setup(const double* b_dev);                       // setUpGenEOPC, geneo.cpp:1672
  int apply(const double* x_dev, double* y_dev);        // applyGenEOPC, geneo.cpp:2051
  int apply_q(const double* x_dev, double* y_dev);      // applyQ, geneo.cpp:1435
  int matmult(const double* x_dev, double* y_dev);      // MatMult(MATIS)
  int solve(const double* b_dev, double* x_dev, KspResult* res);  // KSPSolve counterpart
  int n_owned() const { return (int)owned.size(); }
  const double* x0_dev() const { return d_x0; }
  // results for parity tests
  std::vector<std::vector<double>> eigvals;      // per local subdomain: eigenvalues kept in Z
  std::vector<std::vector<double>> candidates;   // per local subdomain: all converged Ritz values
  std::vector<double> E;                         // dimE x dimE (row-major)
  std::vector<int> ksub_global;                  // realDimE per global subdomain
  std::vector<double> residual_history;
  std::vector<double> tauLoc, gammaLoc;          // per local subdomain (GenEO-2, geneo.cpp:1097-1232)

 private:
  bool is_setup = false;
  int nL = 0, nE = 0, nH = 0;        // local space, ext (= owned + halo), halo sizes
  std::vector<int> suboff;
  bk::Chunks ch;
  bk::Csr neuL, neuE, dirL;          // block-diagonal CSRs (neuE shares rowptr/val with neuL)
  int *d_l2e = nullptr, *d_rt_ptr = nullptr, *d_rt_idx = nullptr, *d_send_idx = nullptr;
  int *d_rv_ptr = nullptr, *d_rv_idx = nullptr, *d_rv_tgt = nullptr;
  int n_rv = 0;
  double *d_D = nullptr, *d_dinv1 = nullptr, *d_dinvN = nullptr;
  double *d_xe = nullptr, *d_ye = nullptr, *d_xL = nullptr, *d_wL = nullptr;
  double *d_cg_r = nullptr, *d_cg_z = nullptr, *d_cg_p = nullptr, *d_cg_q = nullptr, *d_cg_sc = nullptr;
  double *d_t1 = nullptr, *d_t2 = nullptr, *d_t3 = nullptr, *d_x0 = nullptr, *d_scal = nullptr;
  double *d_rvtmp = nullptr;
  // coarse space
  double* d_Z = nullptr;
  int64_t* d_zbase = nullptr;
  int *d_ksub = nullptr, *d_zoff = nullptr, *d_subgid = nullptr;
  std::vector<int> ksub, zoff;
  int kmax = 0, dimE = 0;
  double* d_yE = nullptr;
  double *d_EL = nullptr, *d_ELT = nullptr;   // Cholesky factor of E and its transpose on the device (coarse_solve_local)
  std::vector<double> Efac, EfacT;
  std::vector<int> Epiv;
  bool E_chol = true;
  std::vector<double> h_yE;
  std::vector<double> h_Dscratch;   // partition of unity on the host, reused by the next set-up
  double cheb_lmax = 2.0, cheb_lmax1 = 2.0;
  struct Amg1Pending;
  std::unique_ptr<Amg1Pending> pend1;   // level-1 hierarchy whose host set-up is still running
  int finish_amg1();
  bool eig_only = false;       // a group's temporary PC (eigen_grouped): the set-up stops behind the eigensolve
  std::vector<int> eig_groups; // boundaries of the subdomain groups of this set-up ({0, ns}: one group = the plain path)
  double prepare_secs = 0.0;   // time of setup_prepare (setupTime = both halves)
  double release_secs = 0.0;   // time setup_prepare spent releasing the previous set-up (outside setupTime)
  std::promise<bool>* subs_released = nullptr;   // PC::setup's overlapped mode: fulfilled once setup_prepare no longer reads `subs`
  bool eig_early = false;      // the eigensolves of this set-up have run already (overlapped mode)
  void fill_partition_of_unity();
  int setup_level2_eigen();
  int setup_prepare();         // first half of setup(): everything the eigensolve waits for
  int setup_finish(const double* b_dev);   // second half: level 2 and the bookkeeping
  std::vector<int> plan_eig_groups() const;
  int eigen_grouped();
  std::map<int, void*> cg_graphs;   // HIP graphs of an inner-PCG chunk, by 2 * chunk length + rz parity at its start (local_solve)
  int cg_long_len = 0;         // length of the first chunk of a local solve once the first solve of this set-up is known (0: not yet, -1: never)
  bool cg_graph_failed = false;
  long long cg_chunks = 0;     // chunks issued so far (sampling of direct launches while the in-situ timer runs)
  HostCsr host_neu_cache, host_dir_cache;   // block-diagonal host copies of A_Neu / the level-1 matrix, reused by the next set-up
  AmgDevice* amg1 = nullptr;   // hierarchy of the level-1 (Dirichlet / Robin) block-diagonal matrix (local solves)
  AmgDevice* amgN = nullptr;   // hierarchy of the Neumann block-diagonal matrix (LOBPCG preconditioner)

  int fail(const std::string& msg);
  int build_layout();
  int ensure_dirichlet();
  void make_robin(Sub& s, HostCsr& out) const;
  int setup_level2(const double* b_dev);
  struct EigProblem {
    const bk::Csr* A; const double* As;   // Y = As .* A (As .* X); As may be null
    const bk::Csr* B; const double* Bs;
    AmgDevice* amg;                       // V-cycle of A as preconditioner (null: Chebyshev-Jacobi)
    const double* dinv; double lmax;      // Jacobi scaling and Gershgorin bound for the Chebyshev fallback
    int nev_try;
    const char* label;
    // deflated restart (no -geneo_cut and more eigenvalues below the threshold than one block holds): the iteration runs
    // in the B-orthogonal complement of the locked blocks Y_b (n_L x k_b row-major, B-orthonormal per subdomain, zero
    // columns where a subdomain locked fewer), BY_b = B Y_b; subdomains with skip[s] != 0 are frozen from the start
    struct Locked { const double* Y; const double* BY; int k; };
    const std::vector<Locked>* defl = nullptr;
    const char* skip = nullptr;
    const int* locked_cols = nullptr;     // per subdomain: columns locked so far (bounds the pairs that can still be asked for)
    int seed_off = 0;
    double tol = 0.0;                     // > 0: convergence tolerance of this solve instead of -els2_eps_tol
    // A and B on ONE sliced pattern (bk::spmm_dual): A W and B W of an iteration in one pass over W
    const bk::Csr* dual_pat = nullptr;
    const double *dual_vA = nullptr, *dual_vB = nullptr;
    // A pencil that does not live on the fine rows (the coarse start of eigen_lobpcg: the Galerkin pencil of multigrid
    // level `amg_level`): its row count, the chunks of its subdomain blocks and the blocks' row offsets (ns + 1)
    int rows = -1;
    const bk::Chunks* chunks = nullptr;
    const int* row_off = nullptr;
    int amg_level = 0;                    // the V-cycle starts at this level of `amg`
    const double* X0 = nullptr;           // start block (rows x m row-major, device) instead of the seeded random one
    int* iterations = nullptr;            // where the iteration count goes (default: info.eig_iterations)
    int max_it = 0;                       // > 0: at most this many iterations, and the block is handed back as it stands then
                                          // (a start block for the next level does not have to meet the tolerance)
  };
  int lobpcg_solve(const EigProblem& P, int m, std::vector<double>& lam, double* Xc);
  int eig_targets(int* nev_try) const;
  int eig_block_max() const;
  void local_tau();
  int local_gamma();
  int check_global_spd();
  int check_local_spd(const EigProblem& P, const HostCsr* const* hostB, bool scale_mult);
  int check_local_rank();
  int check_global_rank();
  int eigen_dense_host();
  int eigen_lobpcg();
  int build_E();
  void restrict_to_local(const double* x_owned, double* xL);      // R  (applyLevel1Scatter)
  void prolong_add(const double* wL, double* y_owned);            // sum R^T (applyLevel1Gather)
  void restrict_block(const double* X, double* XL, int w, double* xe);
  void prolong_block(const double* WL, double* Y, int w, double* ye);
  void matmult_block(const double* X, double* Y, int w, double* WL, double* xe);
  void local_solve(double* wL);                                   // [D] M^-1 [D]
  void coarse_solve_local(const double* xL, double* yE);          // yE = E^-1 Z^T x (from xL)
  void allreduce(double* dev, int n);
  double gdot(const double* x, const double* y);
  int solve_cg(const double* b, double* x, KspResult* res);
  int solve_gmres(const double* b, double* x, KspResult* res);
  void free_all();
};

}  // namespace geneo
