// extern "C" entry points of libgeneopc (declared in include/geneo_c.h).
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/geneo_c.h"
#include "core.h"

struct _p_GeneoPC {
  geneo::PC* ctx = nullptr;   // pc->data in PETSc (src/geneo.cpp:2645)
  std::string name, err, optstr;
  // PCSetOperators_GenEO copy (the MATIS A of the one-subdomain-per-rank model)
  bool has_ops = false;
  int nbDOF = 0, nbDOFLoc = 0;
  std::vector<int> map, rowptr, col;
  std::vector<double> val;
  const double* b_dev = nullptr;
};

struct _p_GeneoSpmv {
  bk::Csr a;
};

static std::string g_global_err;

#define GUARD_BEGIN try {
#define GUARD_END(pc)                                \
  }                                                  \
  catch (std::exception & e) {                       \
    if (pc) (pc)->err = e.what();                    \
    else g_global_err = e.what();                    \
    return 1;                                        \
  }

static int pcfail(PC pc, const std::string& m) {
  if (pc) pc->err = m;
  return 1;
}
static int propagate(PC pc, int rc) {
  if (rc && pc && pc->ctx) pc->err = pc->ctx->last_error;
  return rc;
}

extern "C" {

PetscErrorCode PCCreate_GenEO(PC* pc) {
  if (!pc) return 1;
  *pc = new _p_GeneoPC();
  return createGenEOPC(*pc);
}

PetscErrorCode PCGenEOCreateContext(PC pc) { return createGenEOPC(pc); }
PetscErrorCode createGenEOPC(PC pc) {
  if (!pc) return 1;  // "GenEO preconditioner is invalid"
  delete pc->ctx;
  pc->ctx = new geneo::PC();
  pc->name = pc->ctx->opt.name();
  return 0;
}

PetscErrorCode PCDestroy_GenEO(PC* pc) {
  if (!pc || !*pc) return 0;
  GUARD_BEGIN
  delete (*pc)->ctx;
  delete *pc;
  *pc = nullptr;
  bk::alloc_cache_release();      // blocks parked by the caching allocator go back to the device
  GUARD_END((PC) nullptr)
  return 0;
}

static const char* kFlagOptions[] = {"-geneo_cst", "-geneo_no_syl", "-geneo_offload"};
static bool is_flag(const std::string& k) {
  for (const char* f : kFlagOptions)
    if (k == f) return true;
  return false;
}

PetscErrorCode PCGenEOSetOption(PC pc, const char* key, const char* value) {
  if (!pc || !pc->ctx) return 1;
  const std::string k = key ? key : "", v = value ? value : "";
  std::string e = geneo::parse_option(pc->ctx->opt, k, v);
  if (!e.empty()) return pcfail(pc, e);
  e = geneo::validate_options(pc->ctx->opt);
  if (!e.empty()) return pcfail(pc, e);
  pc->name = pc->ctx->opt.name();
  return 0;
}

PetscErrorCode PCSetFromOptions_GenEO(PC pc, int argc, const char* const* argv) {
  if (!pc || !pc->ctx) return 1;
  for (int i = 0; i < argc; ++i) {
    const std::string k = argv[i] ? argv[i] : "";
    if (k.empty() || k[0] != '-') continue;
    if (is_flag(k)) {
      geneo::parse_option(pc->ctx->opt, k, "");
      continue;
    }
    const bool known = k.rfind("-geneo_", 0) == 0 || k.rfind("-els2_", 0) == 0 || k.rfind("-dls1_", 0) == 0 ||
                       k.rfind("-ksp_", 0) == 0 || k.rfind("-amg_", 0) == 0;
    if (!known) continue;
    if (i + 1 >= argc) return pcfail(pc, "invalid option " + k);
    const std::string v = argv[i + 1];
    std::string e = geneo::parse_option(pc->ctx->opt, k, v);
    if (!e.empty()) {
      if (e.rfind("unknown option", 0) == 0) continue;  // forwarded prefix this build does not use
      return pcfail(pc, e);
    }
    ++i;
  }
  std::string e = geneo::validate_options(pc->ctx->opt);
  if (!e.empty()) return pcfail(pc, e);
  pc->name = pc->ctx->opt.name();
  return 0;
}

const char* PCGenEOGetName(PC pc) { return pc ? pc->name.c_str() : ""; }
const char* PCGenEOGetOptionsString(PC pc) {
  if (!pc || !pc->ctx) return "";
  const geneo::Options& o = pc->ctx->opt;
  char buf[1024];
  snprintf(buf, sizeof(buf),
           "lvl1ASM=%d;lvl1RAS=%d;lvl1SRAS=%d;lvl1ORAS=%d;lvl2=%d;hybrid=%d;effHybrid=%d;optim=%.17g;tau=%.17g;"
           "gamma=%.17g;cst=%d;cut=%d;noSyl=%d;offload=%d;eps_tol=%.17g;dls1_rtol=%.17g;dls1_pc=%s;els2_pc=%s;"
           "ksp_type=%s;ksp_rtol=%.17g;ksp_atol=%.17g;ksp_max_it=%d;ksp_restart=%d",
           (int)o.lvl1ASM, (int)o.lvl1RAS, (int)o.lvl1SRAS, (int)o.lvl1ORAS, o.lvl2, (int)o.hybrid, (int)o.effHybrid,
           o.optim, o.tau, o.gamma, (int)o.cst, o.cut, (int)o.noSyl, (int)o.offload, o.eps_tol, o.dls1_rtol,
           o.dls1_pc.c_str(), o.els2_pc.c_str(), o.ksp_type.c_str(), o.ksp_rtol, o.ksp_atol, o.ksp_max_it,
           o.ksp_restart);
  pc->optstr = buf;
  return pc->optstr.c_str();
}
const char* PCGenEOGetError(PC pc) { return pc ? pc->err.c_str() : g_global_err.c_str(); }

const char* usageGenEO_c(void) {
  return "\nusage: GenEO (Domain Decomposition Method) on MI355X\n\n"
         "  -geneo_lvl L1,L2 preconditioner with 2 levels L1 and L2\n"
         "                   L1 = ASM | RAS | SRAS | ORAS | SORAS\n"
         "                   L2 = 0 | 1 | H1 | E1 | 2 | H2 | E2\n"
         "  -geneo_optim A   robin = dirichlet + optim * neumann (ORAS, SORAS; defaults to 0.)\n"
         "  -geneo_tau T     tau threshold (defaults to 0.1)\n"
         "  -geneo_gamma G   gamma threshold (defaults to 10.)\n"
         "  -geneo_cst       do not allow local variations of tau and gamma (GenEO-2)\n"
         "  -geneo_cut C     maximum number of local eigen vectors used to build Z\n"
         "                   (without it every eigenvalue below tau is kept: the LOBPCG block grows up to 64 columns)\n"
         "  -geneo_no_syl    ask for -els2_eps_nev eigenvalues instead (no inertia estimate exists on the GPU path)\n"
         "  -geneo_offload   accepted for compatibility (E is replicated on every GPU)\n"
         "  -geneo_chk F     perform additional checks (F = log | bin | mat; files are text)\n"
         "                     - check partition of unity\n"
         "                     - check matrices are SPD (check.SPD.A.log, check<id>.SPD.<pb>.B.log)\n"
         "                     - check R from Z=QR (check<id>.setup.Z.R, check.setup.ZE2G.R)\n"
         "  -geneo_nicolaides_zero X   the Nicolaides rule takes min(lambda) >= X eps as 'no zero eigenvalue found'\n"
         "                   (1 = the reference's literal test; defaults to 100)\n"
         "  -geneo_eig_group_rows R / -geneo_eig_mem_gb G   memory-bounded set-up: eigensolve the rank's subdomains in\n"
         "                   consecutive groups of at most R local rows / G GiB of basis blocks (default: 35 % of the card)\n"
         "  -geneo_eig_coarse_start R   subdomains of R rows or more (all of a batch): the local eigensolve starts from the Ritz\n"
         "                   vectors of the multigrid level-1 pencil (itself started from level 2, ...) instead of a random\n"
         "                   block (default 750000; 0 = never)\n"
         "  -els2_eps_tol / -els2_eps_nev / -els2_eps_max_it / -els2_eps_block / -els2_pc_type amg|cheb\n"
         "  -els2_cheb_degree / -els2_cheb_ratio\n"
         "  -dls1_ksp_rtol / -dls1_ksp_max_it / -dls1_pc_type amg|jacobi   local solves (batched PCG)\n"
         "  -dls1_amg_precision single|double   storage of the matrices its V-cycle reads (arithmetic and vectors: double)\n"
         "  -dls1_amg_strength T / -els2_amg_strength T   aggregation from level 1 on ties |a_ij| >= T 0.5^l sqrt(a_ii a_jj) only\n"
         "  -amg_coarse_size / -amg_smooth_degree / -amg_smooth_ratio / -amg_max_levels\n"
         "  -dls1_amg_smooth_ratio R / -els2_amg_smooth_ratio R   the smoothing interval [rho / R, 1.1 rho] per hierarchy\n"
         "  -ksp_type cg|gmres -ksp_rtol -ksp_atol -ksp_max_it -ksp_gmres_restart\n\n";
}

PetscErrorCode PCSetOperators_GenEO(PC pc, const GeneoMatIS* A) {
  if (!pc || !A) return 1;
  if (!A->map || !A->local.rowptr || A->local.n != A->nbDOFLoc)
    return pcfail(pc, "GenEO preconditioner needs the A matrix to be of MATIS type");
  pc->has_ops = true;
  pc->nbDOF = A->nbDOF;
  pc->nbDOFLoc = A->nbDOFLoc;
  pc->map.assign(A->map, A->map + A->nbDOFLoc);
  pc->rowptr.assign(A->local.rowptr, A->local.rowptr + A->nbDOFLoc + 1);
  const int nnz = pc->rowptr[A->nbDOFLoc];
  pc->col.assign(A->local.col, A->local.col + nnz);
  pc->val.assign(A->local.val, A->local.val + nnz);
  return 0;
}

PetscErrorCode PCGenEOSetSizes(PC pc, int nbDOF, int nbSubdomainsGlobal) {
  if (!pc || !pc->ctx) return 1;
  pc->ctx->N = nbDOF;
  pc->ctx->nsub_global = nbSubdomainsGlobal;
  return 0;
}

PetscErrorCode PCGenEOAddSubdomain(PC pc, int gid, int n, const int* map, const int* mult, const GeneoCsr* A,
                                   const GeneoCsr* ADir) {
  if (!pc || !pc->ctx) return 1;
  if (!A || A->n != n) return pcfail(pc, "GenEO preconditioner: bad local matrix");
  if (ADir && ADir->n != n) return pcfail(pc, "GenEO preconditioner: bad dirichlet matrix");
  GUARD_BEGIN
  return propagate(pc, pc->ctx->add_subdomain(gid, n, map, mult, A->rowptr, A->col, A->val,
                                              ADir ? ADir->rowptr : nullptr, ADir ? ADir->col : nullptr,
                                              ADir ? ADir->val : nullptr));
  GUARD_END(pc)
}

PetscErrorCode PCGenEOSetupViews(PC pc, const GeneoCsr* pcADirLoc, GeneoIS mults, const GeneoIS* inters) {
  return PCGenEOSetup(pc, pcADirLoc, mults, inters);
}
PetscErrorCode PCGenEOSetup(PC pc, const GeneoCsr* pcADirLoc, GeneoIS mults, const GeneoIS* inters) {
  if (!pc || !pc->ctx) return 1;
  if (!pc->has_ops) return pcfail(pc, "GenEO preconditioner: PCSetOperators_GenEO must be called first");
  if (mults.n != pc->nbDOFLoc) return pcfail(pc, "Mismatch in dof mult size and local size");
  geneo::PC* c = pc->ctx;
  if (c->N == 0) c->N = pc->nbDOF;
  if (c->nsub_global == 0) c->nsub_global = c->size;
  GeneoCsr a{pc->nbDOFLoc, pc->rowptr.data(), pc->col.data(), pc->val.data()};
  if (PetscErrorCode rc = PCGenEOAddSubdomain(pc, c->rank, pc->nbDOFLoc, pc->map.data(), mults.idx, &a, pcADirLoc))
    return rc;
  if (inters) {  // only the emptiness of each list is used, and only by GenEO-2 (src/geneo.cpp:1139-1148)
    std::vector<int> flags(c->nsub_global);
    for (int q = 0; q < c->nsub_global; ++q) flags[q] = inters[q].n > 0;
    return PCGenEOSetIntersect(pc, c->rank, c->nsub_global, flags.data());
  }
  return 0;
}

PetscErrorCode PCGenEOSetIntersect(PC pc, int gid, int nb, const int* nonempty) {
  if (!pc || !pc->ctx) return 1;
  GUARD_BEGIN
  return propagate(pc, pc->ctx->set_intersect(gid, nb, nonempty));
  GUARD_END(pc)
}

PetscErrorCode initGenEOPC_c(PC pc, unsigned int nbDOF, unsigned int nbDOFLoc, const int* map, const GeneoCsr* A,
                             const GeneoCsr* ADir, const double* b_dev, double* x0_dev, const unsigned int* mult) {
  (void)x0_dev;  // fetched with PCGenEOGetX0 after setup
  if (!pc || !pc->ctx) return 1;
  if (!mult) return pcfail(pc, "GenEO preconditioner without DOF multiplicity");
  geneo::PC* c = pc->ctx;
  c->N = (int)nbDOF;
  if (c->nsub_global == 0) c->nsub_global = c->size;
  std::vector<int> m(mult, mult + nbDOFLoc);
  pc->b_dev = b_dev;
  return PCGenEOAddSubdomain(pc, c->rank, (int)nbDOFLoc, map, m.data(), A, ADir);
}

PetscErrorCode PCGenEOSetComm(PC pc, int rank, int size, int n_owned, const int* owned_gid, int n_halo,
                              const int* halo_gid, const int* recv_counts, const int* send_counts,
                              const int* send_idx, GeneoExchangeFn exchange, GeneoAllreduceFn allreduce, void* user,
                              double* send_dev, double* recv_dev, double* red_dev, int red_capacity) {
  if (!pc || !pc->ctx) return 1;
  geneo::PC* c = pc->ctx;
  if (size < 1 || rank < 0 || rank >= size) return pcfail(pc, "GenEO: bad communicator");
  if (size > 1 && (red_capacity < 1 || !red_dev || !exchange || !allreduce))
    return pcfail(pc, "GenEO: communicator needs both callbacks and a reduction buffer of at least one double");
  c->rank = rank;
  c->size = size;
  c->owned.assign(owned_gid, owned_gid + n_owned);
  c->halo_gid.assign(halo_gid, halo_gid + n_halo);
  if (size > 1) {
    c->recv_counts.assign(recv_counts, recv_counts + size);
    c->send_counts.assign(send_counts, send_counts + size);
    int ns = 0;
    for (int q = 0; q < size; ++q) ns += send_counts[q];
    c->send_idx.assign(send_idx, send_idx + ns);
  }
  c->cb_exchange = exchange;
  c->cb_allreduce = allreduce;
  c->cb_user = user;
  c->comm_send = send_dev;
  c->comm_recv = recv_dev;
  c->comm_red = red_dev;
  c->comm_red_cap = red_capacity;
  return 0;
}

PetscErrorCode PCGenEOSetCommWidth(PC pc, int max_width) {
  if (!pc || !pc->ctx) return 1;
  if (max_width < 1) return pcfail(pc, "GenEO: bad halo buffer width");
  pc->ctx->comm_width = max_width;
  return 0;
}

PetscErrorCode PCGenEOSetRHS(PC pc, const double* b_dev) {
  if (!pc) return 1;
  pc->b_dev = b_dev;
  return 0;
}

PetscErrorCode PCSetUp_GenEO(PC pc) {
  if (!pc || !pc->ctx) return 1;
  GUARD_BEGIN
  geneo::PC* c = pc->ctx;
  if (c->nsub_global == 0) c->nsub_global = (int)c->subs.size();
  return propagate(pc, c->setup(pc->b_dev));
  GUARD_END(pc)
}
PetscErrorCode PCApply_GenEO(PC pc, const double* x, double* y) {
  if (!pc || !pc->ctx) return 1;
  GUARD_BEGIN
  return propagate(pc, pc->ctx->apply(x, y));
  GUARD_END(pc)
}
PetscErrorCode PCGenEOApplyQ(PC pc, const double* x, double* y) {
  if (!pc || !pc->ctx) return 1;
  GUARD_BEGIN
  return propagate(pc, pc->ctx->apply_q(x, y));
  GUARD_END(pc)
}
PetscErrorCode MatMult_GenEO(PC pc, const double* x, double* y) {
  if (!pc || !pc->ctx) return 1;
  GUARD_BEGIN
  return propagate(pc, pc->ctx->matmult(x, y));
  GUARD_END(pc)
}
PetscErrorCode PCGenEOGetX0(PC pc, double* x0_dev) {
  if (!pc || !pc->ctx || !pc->ctx->x0_dev()) return 1;
  GUARD_BEGIN
  bk::d2d(x0_dev, pc->ctx->x0_dev(), sizeof(double) * pc->ctx->n_owned());
  GUARD_END(pc)
  return 0;
}
PetscErrorCode KSPSolve_GenEO(PC pc, const double* b, double* x, int* its, double* rnorm, int* reason) {
  if (!pc || !pc->ctx) return 1;
  GUARD_BEGIN
  geneo::KspResult r;
  int rc = pc->ctx->solve(b, x, &r);
  if (its) *its = r.its;
  if (rnorm) *rnorm = r.rnorm;
  if (reason) *reason = r.reason;
  return propagate(pc, rc);
  GUARD_END(pc)
}
int PCGenEOGetResidualHistory(PC pc, double* hist, int cap) {
  if (!pc || !pc->ctx) return 0;
  const auto& h = pc->ctx->residual_history;
  for (int i = 0; i < (int)h.size() && i < cap; ++i) hist[i] = h[i];
  return (int)h.size();
}

PetscErrorCode PCGenEOGetInfo(PC pc, GeneoInfo* o) {
  if (!pc || !pc->ctx || !o) return 1;
  const geneo::Info& i = pc->ctx->info;
  o->estimDimELoc = i.estimDimELoc; o->realDimELoc = i.realDimELoc; o->nicolaidesLoc = i.nicolaidesLoc;
  o->dimE = i.dimE; o->eig_iterations = i.eig_iterations; o->eig_spmm = i.eig_spmm;
  o->dls1_iterations = i.dls1_iterations; o->dls1_solves = i.dls1_solves; o->spmv_calls = i.spmv_calls;
  o->lvl1SetupMinvTimeLoc = i.lvl1SetupMinvTimeLoc; o->lvl2SetupEigTimeLoc = i.lvl2SetupEigTimeLoc;
  o->lvl2SetupZTimeLoc = i.lvl2SetupZTimeLoc; o->lvl2SetupETimeLoc = i.lvl2SetupETimeLoc;
  o->lvl1ApplyTimeLoc = i.lvl1ApplyTimeLoc; o->lvl1ApplyScatterTimeLoc = i.lvl1ApplyScatterTimeLoc;
  o->lvl1ApplyMinvTimeLoc = i.lvl1ApplyMinvTimeLoc; o->lvl1ApplyGatherTimeLoc = i.lvl1ApplyGatherTimeLoc;
  o->lvl1ApplyPrjFSTimeLoc = i.lvl1ApplyPrjFSTimeLoc; o->lvl2ApplyTimeLoc = i.lvl2ApplyTimeLoc;
  o->lvl2ApplyZtTimeLoc = i.lvl2ApplyZtTimeLoc; o->lvl2ApplyEinvTimeLoc = i.lvl2ApplyEinvTimeLoc;
  o->lvl2ApplyZTimeLoc = i.lvl2ApplyZTimeLoc; o->setupTime = i.setupTime; o->solveTime = i.solveTime;
  o->amg_levels = i.amg_levels; o->amg_operator_complexity = i.amg_operator_complexity; o->amgSetupTime = i.amgSetupTime;
  o->nullPivotsLoc = i.nullPivotsLoc;
  o->eigGroups = i.eig_groups;
  o->eigCoarseIterations = i.eig_coarse_iterations;
  return 0;
}
static int copy_out(const std::vector<double>& v, double* out, int cap) {
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
}
int PCGenEOGetEigenvalues(PC pc, int s, double* vals, int cap) {
  if (!pc || !pc->ctx || s < 0 || s >= (int)pc->ctx->eigvals.size()) return -1;
  return copy_out(pc->ctx->eigvals[s], vals, cap);
}
int PCGenEOGetCandidates(PC pc, int s, double* vals, int cap) {
  if (!pc || !pc->ctx || s < 0 || s >= (int)pc->ctx->candidates.size()) return -1;
  return copy_out(pc->ctx->candidates[s], vals, cap);
}
int PCGenEOGetE(PC pc, double* e, int cap) {
  if (!pc || !pc->ctx) return -1;
  copy_out(pc->ctx->E, e, cap);
  return pc->ctx->info.dimE;
}
int PCGenEOGetLocalParams(PC pc, double* tau, double* gamma, int cap) {
  if (!pc || !pc->ctx) return -1;
  const auto& t = pc->ctx->tauLoc;
  const auto& g = pc->ctx->gammaLoc;
  for (int i = 0; i < (int)t.size() && i < cap; ++i) {
    if (tau) tau[i] = t[i];
    if (gamma) gamma[i] = i < (int)g.size() ? g[i] : -1.0;
  }
  return (int)t.size();
}
int PCGenEOGetLocalDims(PC pc, int* k, int cap) {
  if (!pc || !pc->ctx) return -1;
  const auto& v = pc->ctx->ksub_global;
  for (int i = 0; i < (int)v.size() && i < cap; ++i) k[i] = v[i];
  return (int)v.size();
}

// ---- getInput plugin ABI (driver:75-96) -----------------------------------------------------------
// Loads a plugin built for the reference driver (tst/laplacian, tst/heat, tst/graph or a user's own) and
// flattens what it returns.  The C++ signature is the plugin contract itself (driver:81-85).
typedef int (*geneo_get_input_fn)(std::string const& args, unsigned int& nbElem, unsigned int& nbNode,
                                  std::vector<unsigned int>& elemPtr, std::vector<unsigned int>& elemIdx,
                                  std::vector<std::vector<double>>& elemSubMat);
PetscErrorCode GeneoGetLibInput(const char* inpLibA, const char* inpLibArg, GeneoInput* out) {
  if (!inpLibA || !out) return 1;
  memset(out, 0, sizeof(*out));
  void* lib = dlopen(inpLibA, RTLD_LAZY | RTLD_LOCAL);
  if (!lib) {
    g_global_err = std::string("Error: open library KO - ") + dlerror();
    return 1;
  }
  geneo_get_input_fn fn = (geneo_get_input_fn)dlsym(lib, "getInput");
  if (!fn) {
    g_global_err = std::string("Error: get input function from library KO - ") + dlerror();
    dlclose(lib);
    return 1;
  }
  std::string args = inpLibArg ? inpLibArg : "";
  for (auto& c : args)
    if (c == '#') c = ' ';
  unsigned int ne = 0, nn = 0;
  std::vector<unsigned int> ptr, idx;
  std::vector<std::vector<double>> sub;
  int rc = 1;
  try {
    rc = fn(args, ne, nn, ptr, idx, sub);
  } catch (std::exception& e) {
    g_global_err = e.what();
  }
  if (rc != 0 || ptr.size() != (size_t)ne + 1 || sub.size() != ne) {
    if (g_global_err.empty() || rc != 0) g_global_err = "Error: get input data from library KO";
    dlclose(lib);
    return 1;
  }
  size_t tot = 0;
  for (auto& m : sub) tot += m.size();
  out->nbElem = ne;
  out->nbNode = nn;
  out->nIdx = idx.size();
  out->nMat = tot;
  out->elemPtr = (unsigned int*)malloc(sizeof(unsigned int) * (ptr.size() + 1));
  out->elemIdx = (unsigned int*)malloc(sizeof(unsigned int) * (idx.size() + 1));
  out->elemMat = (double*)malloc(sizeof(double) * (tot + 1));
  memcpy(out->elemPtr, ptr.data(), sizeof(unsigned int) * ptr.size());
  memcpy(out->elemIdx, idx.data(), sizeof(unsigned int) * idx.size());
  size_t pos = 0;
  for (auto& m : sub) {
    memcpy(out->elemMat + pos, m.data(), sizeof(double) * m.size());
    pos += m.size();
  }
  dlclose(lib);
  return 0;
}
void GeneoFreeInput(GeneoInput* in) {
  if (!in) return;
  free(in->elemPtr);
  free(in->elemIdx);
  free(in->elemMat);
  memset(in, 0, sizeof(*in));
}

// ---- device helpers ------------------------------------------------------------------------
const char* GeneoBackendName(void) { return bk::name(); }
PetscErrorCode GeneoSetStream(void* s) {
  bk::set_stream(s);
  return 0;
}
int GeneoDeviceCount(void) { return bk::device_count(); }
int GeneoSetDevice(int local_rank) { return bk::set_device(local_rank); }
int GeneoCurrentDevice(void) { return bk::current_device(); }
int GeneoThreadDeviceCheck(void) { return bk::thread_device_check(); }
void GeneoAllocCacheRelease(void) { bk::alloc_cache_release(); }
void* GeneoDeviceAlloc(size_t bytes) {
  try {
    return bk::alloc(bytes);
  } catch (std::exception& e) {
    g_global_err = e.what();
    return nullptr;
  }
}
void GeneoDeviceFree(void* p) { bk::dfree(p); }
PetscErrorCode GeneoH2D(void* d, const void* s, size_t b) {
  GUARD_BEGIN
  bk::h2d(d, s, b);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoD2H(void* d, const void* s, size_t b) {
  GUARD_BEGIN
  bk::d2h(d, s, b);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoDeviceSync(void) {
  GUARD_BEGIN
  bk::sync();
  GUARD_END((PC) nullptr)
  return 0;
}
int GeneoSelfTestMFMA(void) {
  try {
    return bk::selftest_mfma_f64();
  } catch (std::exception& e) {
    g_global_err = e.what();
    return -1;
  }
}
PetscErrorCode GeneoTestAxpby(double* y_dev, const double* x_dev, double a, double b, int n) {
  GUARD_BEGIN
  bk::axpby(y_dev, a, x_dev, b, n);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoSetSpmvKind(int kind) {
  bk::set_spmv_kind(kind);
  return 0;
}
const char* GeneoSpmvKernelName(void) { return bk::spmv_kernel_name(); }
PetscErrorCode GeneoSetMFMA(int enable) {
  bk::set_mfma(enable != 0);
  return 0;
}
PetscErrorCode GeneoSetKernelVariant(const char* name, int value) { return bk::set_variant(name, value) ? 0 : 1; }

// ---- stand-alone kernels --------------------------------------------------------------------
PetscErrorCode GeneoSpmvCreate(const GeneoCsr* a, GeneoSpmv* h) {
  if (!a || !h) return 1;
  GUARD_BEGIN
  *h = new _p_GeneoSpmv();
  (*h)->a = bk::csr_upload(a->n, a->rowptr, a->col, a->val);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoSpmvApply(GeneoSpmv h, const double* x, double* y) {
  if (!h) return 1;
  GUARD_BEGIN
  bk::spmv(h->a, x, y);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoSpmvTime(GeneoSpmv h, const double* x, double* y, int reps, double* ms_avg) {
  if (!h || reps < 1) return 1;
  GUARD_BEGIN
  void* e0 = bk::event_create();
  void* e1 = bk::event_create();
  bk::spmv(h->a, x, y);  // warm
  bk::event_record(e0);
  for (int i = 0; i < reps; ++i) bk::spmv(h->a, x, y);
  bk::event_record(e1);
  const float ms = bk::event_elapsed_ms(e0, e1);
  bk::event_destroy(e0);
  bk::event_destroy(e1);
  if (ms_avg) *ms_avg = (double)ms / reps;
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoSpmvProfileStart(int every, double min_bytes) {
  bk::spmv_profile_start(every, min_bytes);
  return 0;
}
PetscErrorCode GeneoSpmvProfileStop(double* ms_sum, double* bytes_sum, long long* nsampled, long long* nlaunch) {
  GUARD_BEGIN
  bk::spmv_profile_stop(ms_sum, bytes_sum, nsampled, nlaunch);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoDeviceMemInfo(double* live, double* live_peak, double* footprint_peak, double* cached, double* dev_free,
                                  double* dev_total, int reset_peaks) {
  GUARD_BEGIN
  bk::mem_info(live, live_peak, footprint_peak, cached, dev_free, dev_total, reset_peaks != 0);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoKernelProfileStart(int every, double spmv_min_bytes) {
  GUARD_BEGIN
  bk::kernel_profile_start(every, spmv_min_bytes);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoKernelProfileStop(void) {
  GUARD_BEGIN
  bk::kernel_profile_stop();
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoKernelProfileGet(int kernel_class, double* ms_sum, double* bytes_sum, double* flops_sum,
                                     long long* nsampled, long long* nlaunch) {
  GUARD_BEGIN
  bk::kernel_profile_get(kernel_class, ms_sum, bytes_sum, flops_sum, nsampled, nlaunch);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoSpmvDestroy(GeneoSpmv* h) {
  if (!h || !*h) return 0;
  bk::csr_free((*h)->a);
  delete *h;
  *h = nullptr;
  return 0;
}
PetscErrorCode GeneoSpmmApply(GeneoSpmv h, const double* X, double* Y, int m, const double* pre, const double* post) {
  if (!h) return 1;
  GUARD_BEGIN
  bk::spmm_strided(h->a, X, m, Y, m, m, pre, post);
  GUARD_END((PC) nullptr)
  return 0;
}

// strided blocks + HIP-event timing of `reps` back-to-back launches (reps <= 0: one untimed launch)
PetscErrorCode GeneoSpmmTime(GeneoSpmv h, const double* X, int ldx, double* Y, int ldy, int m, const double* pre,
                             const double* post, int reps, double* ms_avg) {
  if (!h) return 1;
  GUARD_BEGIN
  bk::spmm_strided(h->a, X, ldx, Y, ldy, m, pre, post);
  if (reps > 0) {
    void* e0 = bk::event_create();
    void* e1 = bk::event_create();
    bk::event_record(e0);
    for (int i = 0; i < reps; ++i) bk::spmm_strided(h->a, X, ldx, Y, ldy, m, pre, post);
    bk::event_record(e1);
    const float ms = bk::event_elapsed_ms(e0, e1);
    bk::event_destroy(e0);
    bk::event_destroy(e1);
    if (ms_avg) *ms_avg = (double)ms / reps;
  }
  GUARD_END((PC) nullptr)
  return 0;
}

PetscErrorCode GeneoSpmmFused(GeneoSpmv h, int epi, const double* X, double* Y, int m, const double* B, double* Z,
                              const double* dinv, double w) {
  if (!h) return 1;
  GUARD_BEGIN
  bk::spmm_fused(h->a, epi, X, m, Y, m, m, B, m, Z, m, dinv, w);
  GUARD_END((PC) nullptr)
  return 0;
}

// the same single-vector launches reading the single-precision companion of the matrix (built on first use); epi 0: Y = A X
// Y1 = B X and Y2 = A X in one pass over X, B laid out on A's sliced pattern (bk::sell_values_on + bk::spmm_dual: LOBPCG's
// A W / B W pass).  Returns 2 when pattern(B) is not contained in pattern(A) or A is not on the sliced path.
PetscErrorCode GeneoSpmmDualTest(GeneoSpmv a, GeneoSpmv b, const double* X, int ldx, double* Y1, double* Y2, int ldy, int m) {
  if (!a || !b) return 1;
  GUARD_BEGIN
  if (!bk::spmm_dual_available(a->a, m)) return 2;
  double* v = bk::sell_values_on(a->a, b->a);
  if (!v) return 2;
  bk::spmm_dual(a->a, v, a->a.sl_val, X, ldx, Y1, Y2, ldy, m);
  bk::sync();
  bk::dfree(v);
  GUARD_END((PC) nullptr)
  return 0;
}
// R = mask .* (A X - B X diag(lam)) per subdomain (suboff: nsub + 1 first rows; lam, mask: nsub x m host arrays), both
// products in one pass over X and neither written (bk::spmm_dual_residual: the residual block of LOBPCG's lean iteration)
PetscErrorCode GeneoSpmmDualResidualTest(GeneoSpmv a, GeneoSpmv b, const double* X, int ldx, double* R, int ldr, int m,
                                         int nsub, const int* suboff, const double* lam, const double* mask) {
  if (!a || !b) return 1;
  GUARD_BEGIN
  if (!bk::spmm_dual_available(a->a, m)) return 2;
  double* v = bk::sell_values_on(a->a, b->a);
  if (!v) return 2;
  bk::Chunks c = bk::chunks_upload(nsub, suboff);
  double* dl = (double*)bk::alloc(sizeof(double) * (size_t)nsub * m);
  double* dm = (double*)bk::alloc(sizeof(double) * (size_t)nsub * m);
  bk::h2d(dl, lam, sizeof(double) * (size_t)nsub * m);
  bk::h2d(dm, mask, sizeof(double) * (size_t)nsub * m);
  bk::spmm_dual_residual(a->a, a->a.sl_val, v, X, ldx, R, ldr, m, c, dl, dm);
  bk::sync();
  bk::dfree(v); bk::dfree(dl); bk::dfree(dm);
  bk::chunks_free(c);
  GUARD_END((PC) nullptr)
  return 0;
}
PetscErrorCode GeneoSpmvFusedSingle(GeneoSpmv h, int epi, const double* X, double* Y, const double* B, double* Z,
                                    const double* dinv, double w) {
  if (!h) return 1;
  GUARD_BEGIN
  if (!bk::csr_has_lp(h->a) && !bk::csr_make_lp(h->a))
    throw std::runtime_error("no single-precision companion for this matrix (ragged or long rows: not on the sliced path)");
  if (epi == 0) bk::spmv_lp(h->a, X, Y);
  else bk::spmv_fused_lp(h->a, epi, X, Y, B, Z, dinv, w);
  GUARD_END((PC) nullptr)
  return 0;
}

// test hook for the device sparse products: op 0: C = A B, op 1: C = A^T (B ignored).  Returns nnz(C) (-1: a row
// exceeded the kernels' capacity, -2: error); fills the outputs when cap >= nnz (rowptr_out has C's rows + 1 entries).
long long GeneoTestSparseProduct(int op, const GeneoCsr* A, const GeneoCsr* B, int ncols, int* rowptr_out, int* col_out,
                                 double* val_out, long long cap) {
  try {
    bk::Csr a = bk::csr_upload(A->n, A->rowptr, A->col, A->val);
    bk::Csr b;
    if (op == 0) b = bk::csr_upload(B->n, B->rowptr, B->col, B->val);
    bool ok = true;
    bk::Csr c = (op == 0) ? bk::spgemm(a, b, ncols, &ok) : bk::transpose(a, ncols, &ok);
    long long nnz = ok ? (long long)c.nnz : -1;
    if (ok && cap >= nnz && rowptr_out) bk::csr_download(c, rowptr_out, col_out, val_out);
    bk::csr_free(a);
    if (op == 0) bk::csr_free(b);
    if (ok) bk::csr_free(c);
    return nnz;
  } catch (std::exception& e) {
    g_global_err = e.what();
    return -2;
  }
}

// threshold (chunks per subdomain) above which the per-subdomain reductions take their cooperative forms; returns the
// previous value.  Set it BEFORE creating the PC whose solves it should govern (captured HIP graphs keep their launches).
int GeneoSetParReduceMin(int chunks) {
  const int old = bk::get_par_reduce_min();
  bk::set_par_reduce_min(chunks);
  return old;
}
// test hook of the fused LOBPCG update (m = 32): host arrays in, host arrays out
PetscErrorCode GeneoTestLobpcgUpdate(int nsub, const int* suboff, const double* S, const double* AS, const double* BS,
                                     const double* C, const double* keep, const double* lam, const double* mask,
                                     double* T, double* AT, double* BT, double* R) {
  GUARD_BEGIN
  if (!bk::lobpcg_update32_available()) throw std::runtime_error("fused LOBPCG update unavailable (MFMA off)");
  const int n = suboff[nsub];
  bk::Chunks c = bk::chunks_upload(nsub, suboff);
  auto up = [](const double* h, size_t k) {
    double* d = (double*)bk::alloc(sizeof(double) * std::max<size_t>(1, k));
    if (h) bk::h2d(d, h, sizeof(double) * k);
    return d;
  };
  const size_t nb = (size_t)n * 96;
  if (!AS) {   // the basis-only form of the lean iteration: T = [X' P'] from S (AS, BS, lam, mask, AT, BT, R unused)
    double *dS = up(S, nb), *dC = up(C, (size_t)nsub * 96 * 64), *dk = up(keep, (size_t)nsub * 32), *dT = up(nullptr, nb);
    bk::lobpcg_update32_basis(c, dS, dC, dk, dT);
    bk::d2h(T, dT, sizeof(double) * nb);
    for (double* d : {dS, dC, dk, dT}) bk::dfree(d);
    bk::chunks_free(c);
    return 0;
  }
  double *dS = up(S, nb), *dAS = up(AS, nb), *dBS = up(BS, nb), *dC = up(C, (size_t)nsub * 96 * 64);
  double *dk = up(keep, (size_t)nsub * 32), *dl = up(lam, (size_t)nsub * 32), *dm = up(mask, (size_t)nsub * 32);
  double *dT = up(nullptr, nb), *dAT = up(nullptr, nb), *dBT = up(nullptr, nb), *dR = up(nullptr, (size_t)n * 32);
  bk::lobpcg_update32(c, dS, dAS, dBS, dC, dk, dl, dm, dT, dAT, dBT, dR);
  bk::d2h(T, dT, sizeof(double) * nb);
  bk::d2h(AT, dAT, sizeof(double) * nb);
  bk::d2h(BT, dBT, sizeof(double) * nb);
  bk::d2h(R, dR, sizeof(double) * (size_t)n * 32);
  for (double* d : {dS, dAS, dBS, dC, dk, dl, dm, dT, dAT, dBT, dR}) bk::dfree(d);
  bk::chunks_free(c);
  GUARD_END((PC) nullptr)
  return 0;
}

PetscErrorCode GeneoBlockKernel(int kind, int nsub, const int* suboff, const double* S, int p, const double* TC, int q,
                                double* out, int reps, double* ms_avg) {
  GUARD_BEGIN
  const int n = suboff[nsub];
  bk::Chunks c = bk::chunks_upload(nsub, suboff);
  double* dS = (double*)bk::alloc(sizeof(double) * (size_t)n * p);
  bk::h2d(dS, S, sizeof(double) * (size_t)n * p);
  void* e0 = bk::event_create();
  void* e1 = bk::event_create();
  if (kind == 0 || kind == 2) {
    // kind 2: the left operand as TWO strided views (columns [0, p/2) and [p/2, p) of S: what LOBPCG passes as A W, B W)
    double* dT = (double*)bk::alloc(sizeof(double) * (size_t)n * q);
    double* dG = (double*)bk::alloc(sizeof(double) * (size_t)nsub * p * q);
    bk::h2d(dT, TC, sizeof(double) * (size_t)n * q);
    auto run = [&]() {
      if (kind == 0) bk::gram(c, dS, p, p, dT, q, q, dG);
      else bk::gram2(c, dS, p, p / 2, dS + p / 2, p, p - p / 2, dT, q, q, dG);
    };
    run();
    if (reps > 0) {
      bk::event_record(e0);
      for (int i = 0; i < reps; ++i) run();
      bk::event_record(e1);
      if (ms_avg) *ms_avg = bk::event_elapsed_ms(e0, e1) / reps;
    }
    bk::d2h(out, dG, sizeof(double) * (size_t)nsub * p * q);
    bk::dfree(dT);
    bk::dfree(dG);
  } else {
    double* dC = (double*)bk::alloc(sizeof(double) * (size_t)nsub * p * q);
    double* dY = (double*)bk::alloc(sizeof(double) * (size_t)n * q);
    bk::h2d(dC, TC, sizeof(double) * (size_t)nsub * p * q);
    bk::block_mul(c, dS, p, p, dC, q, dY, q, false);
    if (reps > 0) {
      bk::event_record(e0);
      for (int i = 0; i < reps; ++i) bk::block_mul(c, dS, p, p, dC, q, dY, q, false);
      bk::event_record(e1);
      if (ms_avg) *ms_avg = bk::event_elapsed_ms(e0, e1) / reps;
    }
    bk::d2h(out, dY, sizeof(double) * (size_t)n * q);
    bk::dfree(dC);
    bk::dfree(dY);
  }
  bk::dfree(dS);
  bk::chunks_free(c);
  bk::event_destroy(e0);
  bk::event_destroy(e1);
  GUARD_END((PC) nullptr)
  return 0;
}

}  // extern "C"
