// geneo_petsc_adapter.cpp -- PETSc-side adapter for libgeneopc (the MI355X-native GenEO preconditioner).
//
// Takes the place of src/geneo.cpp in the reference tree (src/CMakeLists.txt:10 builds libgeneopc.a from it) and links
// -lgeneopc from this repository.  It defines, with PETSc's own types, everything the reference's driver binds:
//     createGenEOPC(PC)                         hdr/geneo_c.h:9      (PCRegister callback, src/geneo4PETSc.cpp:1331)
//     PCGenEOSetup(PC, Mat, IS, IS*)            hdr/geneo_c.h:10
//     initGenEOPC(PC&, ..., Mat, Mat, Vec, Vec, vector<unsigned>*, ...)   hdr/geneo.hpp:30-35  (driver:1346)
//     usageGenEO(bool)                          hdr/geneo.hpp:41     (driver:1566)
// and it puts a `geneoContext` -- the REFERENCE's own class, hdr/geneo.hpp:46-138, included from the reference tree, not
// re-declared here -- at pc->data, because the driver reads its public members directly (driver:928-989 parameters and
// counters, driver:1123-1225 timers).  The library handle rides behind it (struct Ctx : geneoContext).
// The parameters are refreshed from the library after PCSetFromOptions, the counters / timers after set-up and after
// every apply, so those driver lines print real values.
//
// Several MPI ranks: the MATIS local-to-global map and PETSc's row ownership give the halo plan (who owns the DOFs I
// overlap, who overlaps mine: one MPI_Alltoall + one MPI_Alltoallv at set-up); the data path then runs over the
// library's C++ RCCL transport (PCGenEOSetCommRccl; the 128-byte unique id is broadcast with MPI_Bcast) -- the
// counterpart of the reference's VecScatter (geneo.cpp:1850-1883).
//
// Compile-checked in this repository against a declaration-only PETSc stub (tests/test_adapter.py); building it for
// real needs PETSc >= 3.10 with 32-bit PetscInt and real double PetscScalar (what the reference requires):
//   mpicxx -std=c++11 -I$PETSC_DIR/include -I<reference>/hdr -c <this repo>/adapters/geneo_petsc_adapter.cpp
#include <geneo.hpp>                       // the reference's hdr/geneo.hpp: petsc.h, pcimpl.h, class geneoContext, prototypes

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define GENEO_HAVE_PETSC                   // PETSc's PC / PetscErrorCode stay PETSc's; the library handle is GeneoPC
#include "../include/geneo_c.h"            // the C ABI of libgeneopc (this repository)

namespace {

struct Ctx : public geneoContext {          // pc->data points HERE; the geneoContext base sits at offset 0
  GeneoPC h = NULL;
  GeneoRccl rccl = NULL;
  double *xd = NULL, *yd = NULL;            // device copies of the owned part of the Vec arguments
  PetscInt cap = 0;
  Mat aglob = NULL, *adir_sub = NULL;       // assembled A and its local Dirichlet block (when the caller gave none)
  bool built = false;
};

Ctx* ctx_of(PC pc) { return static_cast<Ctx*>(static_cast<geneoContext*>(pc->data)); }

PetscErrorCode fail(Ctx* c, const char* what) {
  SETERRQ2(PETSC_COMM_SELF, PETSC_ERR_LIB, "GenEO (libgeneopc) %s: %s", what, c && c->h ? PCGenEOGetError(c->h) : "no context");
}

// "key=value;..." of PCGenEOGetOptionsString
double opt_value(const std::string& s, const char* key) {
  const std::string k = std::string(key) + "=";
  size_t p = s.find(k);
  while (p != std::string::npos && p > 0 && s[p - 1] != ';') p = s.find(k, p + 1);
  return p == std::string::npos ? 0.0 : atof(s.c_str() + p + k.size());
}

// the public parameters of geneoContext (hdr/geneo.hpp:52-66), as the library parsed them
void refresh_parameters(Ctx* c) {
  const std::string s = PCGenEOGetOptionsString(c->h);
  c->name = PCGenEOGetName(c->h);
  c->lvl1ASM = opt_value(s, "lvl1ASM") != 0; c->lvl1RAS = opt_value(s, "lvl1RAS") != 0;
  c->lvl1SRAS = opt_value(s, "lvl1SRAS") != 0; c->lvl1ORAS = opt_value(s, "lvl1ORAS") != 0;
  c->lvl2 = (int)opt_value(s, "lvl2");
  c->hybrid = opt_value(s, "hybrid") != 0; c->effHybrid = opt_value(s, "effHybrid") != 0;
  c->optim = opt_value(s, "optim"); c->tau = opt_value(s, "tau"); c->gamma = opt_value(s, "gamma");
  c->cst = opt_value(s, "cst") != 0; c->cut = (int)opt_value(s, "cut");
  c->noSyl = opt_value(s, "noSyl") != 0; c->offload = opt_value(s, "offload") != 0;
  c->infoL2 = "lobpcg cholesky";            // what stands where the reference prints its EPS / coarse solver types (driver:968)
}

// the public counters and timers (hdr/geneo.hpp:96-123)
PetscErrorCode refresh_info(Ctx* c) {
  GeneoInfo i;
  if (PCGenEOGetInfo(c->h, &i)) return fail(c, "PCGenEOGetInfo");
  c->estimDimELoc = i.estimDimELoc; c->realDimELoc = i.realDimELoc; c->nicolaidesLoc = i.nicolaidesLoc;
  c->lvl1SetupMinvTimeLoc = i.lvl1SetupMinvTimeLoc; c->lvl2SetupEigTimeLoc = i.lvl2SetupEigTimeLoc;
  c->lvl2SetupTauEigTimeLoc = i.lvl2SetupEigTimeLoc;
  c->lvl2SetupZTimeLoc = i.lvl2SetupZTimeLoc; c->lvl2SetupETimeLoc = i.lvl2SetupETimeLoc;
  c->lvl1ApplyTimeLoc = i.lvl1ApplyTimeLoc; c->lvl1ApplyScatterTimeLoc = i.lvl1ApplyScatterTimeLoc;
  c->lvl1ApplyMinvTimeLoc = i.lvl1ApplyMinvTimeLoc; c->lvl1ApplyGatherTimeLoc = i.lvl1ApplyGatherTimeLoc;
  c->lvl1ApplyPrjFSTimeLoc = i.lvl1ApplyPrjFSTimeLoc; c->lvl2ApplyTimeLoc = i.lvl2ApplyTimeLoc;
  c->lvl2ApplyZtTimeLoc = i.lvl2ApplyZtTimeLoc; c->lvl2ApplyEinvTimeLoc = i.lvl2ApplyEinvTimeLoc;
  c->lvl2ApplyZTimeLoc = i.lvl2ApplyZTimeLoc;
  return 0;
}

PetscErrorCode csr_of(Mat seqaij, GeneoCsr* v) {            // zero-copy view of a SEQAIJ matrix (32-bit PetscInt)
  const PetscInt *ia, *ja; PetscInt n; PetscBool ok; PetscScalar* a; PetscErrorCode ierr;
  ierr = MatGetRowIJ(seqaij, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &ok); CHKERRQ(ierr);
  if (!ok) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "GenEO: local matrix is not SEQAIJ");
  ierr = MatSeqAIJGetArray(seqaij, &a); CHKERRQ(ierr);
  v->n = (int)n; v->rowptr = (const int*)ia; v->col = (const int*)ja; v->val = a;
  return 0;
}

// Halo plan of this rank from the MATIS map and PETSc's contiguous row ownership [rs, re): the DOFs I read but do not
// own, grouped by owner (ascending global id inside a group), and what every other rank reads from me.
PetscErrorCode halo_plan(MPI_Comm comm, PetscInt rs, PetscInt re, const PetscInt* l2g, PetscInt n,
                         std::vector<int>& halo, std::vector<int>& recv_counts, std::vector<int>& send_counts,
                         std::vector<int>& send_idx) {
  PetscMPIInt size, rank; PetscErrorCode ierr;
  ierr = MPI_Comm_size(comm, &size); CHKERRQ(ierr);
  ierr = MPI_Comm_rank(comm, &rank); CHKERRQ(ierr);
  std::vector<int> starts(size + 1, 0), my(1, (int)rs);
  ierr = MPI_Allgather(my.data(), 1, MPI_INT, starts.data(), 1, MPI_INT, comm); CHKERRQ(ierr);
  int nglob = (int)re;
  ierr = MPI_Allreduce(MPI_IN_PLACE, &nglob, 1, MPI_INT, MPI_MAX, comm); CHKERRQ(ierr);
  starts[size] = nglob;
  halo.clear();
  for (PetscInt i = 0; i < n; ++i)
    if (l2g[i] < rs || l2g[i] >= re) halo.push_back((int)l2g[i]);
  std::sort(halo.begin(), halo.end());                    // ranges ascend with the rank: sorted ids are grouped by owner
  halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
  recv_counts.assign(size, 0);
  for (int g : halo) recv_counts[(int)(std::upper_bound(starts.begin(), starts.end(), g) - starts.begin()) - 1]++;
  send_counts.assign(size, 0);
  ierr = MPI_Alltoall(recv_counts.data(), 1, MPI_INT, send_counts.data(), 1, MPI_INT, comm); CHKERRQ(ierr);
  std::vector<int> rdis(size + 1, 0), sdis(size + 1, 0);
  for (int q = 0; q < size; ++q) { rdis[q + 1] = rdis[q] + recv_counts[q]; sdis[q + 1] = sdis[q] + send_counts[q]; }
  send_idx.assign(std::max(1, sdis[size]), 0);
  ierr = MPI_Alltoallv(halo.data(), recv_counts.data(), rdis.data(), MPI_INT, send_idx.data(), send_counts.data(),
                       sdis.data(), MPI_INT, comm); CHKERRQ(ierr);
  send_idx.resize(sdis[size]);
  for (int& g : send_idx) g -= (int)rs;                  // owned-local indices
  return 0;
}

PetscErrorCode to_device(Ctx* c, Vec v, double* d) {
  const PetscScalar* a; PetscInt n; PetscErrorCode ierr;
  ierr = VecGetLocalSize(v, &n); CHKERRQ(ierr);
  ierr = VecGetArrayRead(v, &a); CHKERRQ(ierr);
  if (GeneoH2D(d, a, sizeof(double) * (size_t)n)) return fail(c, "host to device copy");
  ierr = VecRestoreArrayRead(v, &a); CHKERRQ(ierr);
  return 0;
}
PetscErrorCode from_device(Ctx* c, Vec v, const double* d) {
  PetscScalar* a; PetscInt n; PetscErrorCode ierr;
  ierr = VecGetLocalSize(v, &n); CHKERRQ(ierr);
  ierr = VecGetArray(v, &a); CHKERRQ(ierr);
  if (GeneoD2H(a, d, sizeof(double) * (size_t)n)) return fail(c, "device to host copy");
  ierr = VecRestoreArray(v, &a); CHKERRQ(ierr);
  return 0;
}
PetscErrorCode reserve(Ctx* c, PetscInt n) {
  if (n <= c->cap) return 0;
  GeneoDeviceFree(c->xd); GeneoDeviceFree(c->yd);
  c->xd = (double*)GeneoDeviceAlloc(sizeof(double) * (size_t)n);
  c->yd = (double*)GeneoDeviceAlloc(sizeof(double) * (size_t)n);
  c->cap = n;
  return (c->xd && c->yd) ? 0 : fail(c, "device allocation");
}

// ops->setup (setUpGenEOPC, geneo.cpp:1672): hand the operators to the library, wire the transport, set up
PetscErrorCode setup(PC pc) {
  Ctx* c = ctx_of(pc);
  PetscErrorCode ierr;
  if (!c->pcA) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_NULL, "GenEO preconditioner without A matrix");   // geneo.cpp:1678
  PetscBool isMatIS = PETSC_FALSE;
  ierr = PetscObjectTypeCompare((PetscObject)c->pcA, MATIS, &isMatIS); CHKERRQ(ierr);
  if (!isMatIS) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_WRONG, "GenEO preconditioner A must be MatIS");   // geneo.cpp:1681
  if (!c->dofIdxMultLoc) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_NULL, "GenEO preconditioner without DOF multiplicities");
  if (!c->built) {
    PetscMPIInt rank, size;
    ierr = MPI_Comm_rank(PETSC_COMM_WORLD, &rank); CHKERRQ(ierr);
    ierr = MPI_Comm_size(PETSC_COMM_WORLD, &size); CHKERRQ(ierr);
    Mat aloc;
    GeneoCsr neu, dir;
    ierr = MatISGetLocalMat(c->pcA, &aloc); CHKERRQ(ierr);
    ierr = csr_of(aloc, &neu); CHKERRQ(ierr);
    const PetscInt* l2g; PetscInt n;
    ierr = ISLocalToGlobalMappingGetSize(c->pcMap, &n); CHKERRQ(ierr);
    ierr = ISLocalToGlobalMappingGetIndices(c->pcMap, &l2g); CHKERRQ(ierr);
    Mat adir = c->pcADirLoc;
    if (!adir) {   // A_Dir = the local block of the assembled A (geneo.cpp:1692-1705: MatConvert + MatCreateSubMatrices)
      if (!c->pcIS) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_NULL, "GenEO preconditioner without dirichlet matrix nor DOF list");
      ierr = MatConvert(c->pcA, MATAIJ, MAT_INITIAL_MATRIX, &c->aglob); CHKERRQ(ierr);
      ierr = MatCreateSubMatrices(c->aglob, 1, &c->pcIS, &c->pcIS, MAT_INITIAL_MATRIX, &c->adir_sub); CHKERRQ(ierr);
      adir = c->adir_sub[0];
    }
    ierr = csr_of(adir, &dir); CHKERRQ(ierr);
    if (PCGenEOSetSizes(c->h, (int)c->nbDOF, (int)size)) return fail(c, "PCGenEOSetSizes");
    PetscInt rs, re;
    ierr = MatGetOwnershipRange(c->pcA, &rs, &re); CHKERRQ(ierr);
    if (size > 1) {
      std::vector<int> owned((size_t)(re - rs)), halo, rc, sc, si;
      for (PetscInt g = rs; g < re; ++g) owned[(size_t)(g - rs)] = (int)g;
      ierr = halo_plan(PETSC_COMM_WORLD, rs, re, l2g, n, halo, rc, sc, si); CHKERRQ(ierr);
      // Collective-safe bootstrap: every step is agreed on by all ranks before anyone raises, so that a rank failing
      // alone (RCCL not loadable, communicator refused) never leaves the others waiting in a collective.
      char id[129];
      id[128] = (rank == 0 && GeneoRcclUniqueId(id)) ? 0 : 1;          // byte 128: rank 0 has an id
      ierr = MPI_Bcast(id, 129, MPI_BYTE, 0, PETSC_COMM_WORLD); CHKERRQ(ierr);
      if (!id[128]) SETERRQ1(PETSC_COMM_WORLD, PETSC_ERR_LIB, "GenEO: rank 0 could not create the RCCL unique id: %s", rank == 0 ? GeneoRcclGetError() : "see rank 0");
      int made = GeneoRcclCreate(id, rank, size, &c->rccl) ? 0 : 1, all_made = 0;
      ierr = MPI_Allreduce(&made, &all_made, 1, MPI_INT, MPI_MIN, PETSC_COMM_WORLD); CHKERRQ(ierr);
      if (!all_made) {
        if (made) GeneoRcclDestroy(&c->rccl);
        SETERRQ1(PETSC_COMM_WORLD, PETSC_ERR_LIB, "GenEO: RCCL communicator creation failed on at least one rank: %s", made ? "another rank" : GeneoRcclGetError());
      }
      if (PCGenEOSetCommRccl(c->h, c->rccl, (int)owned.size(), owned.data(), (int)halo.size(), halo.data(), rc.data(),
                             sc.data(), si.data(), 32))
        SETERRQ1(PETSC_COMM_SELF, PETSC_ERR_LIB, "GenEO: %s", GeneoRcclGetError());
    }
    std::vector<int> mult(c->dofIdxMultLoc->begin(), c->dofIdxMultLoc->end());
    if ((PetscInt)mult.size() != n) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_SIZ, "GenEO preconditioner bad DOF multiplicity");
    if (PCGenEOAddSubdomain(c->h, (int)rank, (int)n, (const int*)l2g, mult.data(), &neu, &dir)) return fail(c, "PCGenEOAddSubdomain");
    if (c->intersectLoc) {   // only the emptiness of each list is used, and only by GenEO-2 (geneo.cpp:1139-1148)
      std::vector<int> nonempty(c->intersectLoc->size());
      for (size_t q = 0; q < nonempty.size(); ++q) nonempty[q] = (*c->intersectLoc)[q].empty() ? 0 : 1;
      if (PCGenEOSetIntersect(c->h, (int)rank, (int)nonempty.size(), nonempty.data())) return fail(c, "PCGenEOSetIntersect");
    }
    ierr = ISLocalToGlobalMappingRestoreIndices(c->pcMap, &l2g); CHKERRQ(ierr);
    ierr = reserve(c, re - rs); CHKERRQ(ierr);
    c->built = true;
  }
  if (c->pcB) {   // right-hand side: the efficient hybrid derives the initial guess Q b from it (geneo.cpp:1601-1604)
    ierr = to_device(c, c->pcB, c->xd); CHKERRQ(ierr);
    if (PCGenEOSetRHS(c->h, c->xd)) return fail(c, "PCGenEOSetRHS");
  }
  if (PCSetUp_GenEO(c->h)) return fail(c, "set-up");
  if (c->pcX0) {  // written by the set-up: Q b for the efficient hybrid, zero otherwise (geneo.cpp:1602 / :1606)
    if (PCGenEOGetX0(c->h, c->yd)) return fail(c, "PCGenEOGetX0");
    ierr = from_device(c, c->pcX0, c->yd); CHKERRQ(ierr);
  }
  return refresh_info(c);
}

// ops->apply (applyGenEOPC, geneo.cpp:2051).  Host Vecs: two PCIe copies of the owned part per application; with a
// HIP-enabled PETSc pass the device arrays (VecHIPGetArrayRead / VecHIPGetArrayWrite) straight to PCApply_GenEO.
PetscErrorCode apply(PC pc, Vec x, Vec y) {
  Ctx* c = ctx_of(pc);
  PetscInt n; PetscErrorCode ierr;
  ierr = VecGetLocalSize(x, &n); CHKERRQ(ierr);
  ierr = reserve(c, n); CHKERRQ(ierr);
  ierr = to_device(c, x, c->xd); CHKERRQ(ierr);
  if (PCApply_GenEO(c->h, c->xd, c->yd)) return fail(c, "apply");
  ierr = from_device(c, y, c->yd); CHKERRQ(ierr);
  return refresh_info(c);
}

// ops->destroy (destroyGenEOPC, geneo.cpp:2180-2243): owned objects only; pcA, pcMap, the vectors of multiplicities
// and intersections are borrowed (geneo.cpp:2221-2230)
PetscErrorCode destroy(PC pc) {
  Ctx* c = ctx_of(pc);
  PetscErrorCode ierr;
  if (!c) return 0;
  GeneoDeviceFree(c->xd); GeneoDeviceFree(c->yd);
  if (c->h) PCDestroy_GenEO(&c->h);
  if (c->rccl) GeneoRcclDestroy(&c->rccl);
  if (c->adir_sub) { ierr = MatDestroySubMatrices(1, &c->adir_sub); CHKERRQ(ierr); }
  if (c->aglob) { ierr = MatDestroy(&c->aglob); CHKERRQ(ierr); }
  if (c->pcADirLoc) { ierr = MatDestroy(&c->pcADirLoc); CHKERRQ(ierr); }
  if (c->pcB) { ierr = VecDestroy(&c->pcB); CHKERRQ(ierr); }
  if (c->pcX0) { ierr = VecDestroy(&c->pcX0); CHKERRQ(ierr); }
  if (c->pcIS) { ierr = ISDestroy(&c->pcIS); CHKERRQ(ierr); }
  if (c->pcKSPL1Loc) { ierr = KSPDestroy(&c->pcKSPL1Loc); CHKERRQ(ierr); }
  delete c;
  pc->data = NULL;
  return 0;
}

// ops->setfromoptions (setUpGenEOPCFromOptions, geneo.cpp:2329): the library parses the same -geneo_* spellings
PetscErrorCode setfromoptions(PetscOptionItems*, PC pc) {
  Ctx* c = ctx_of(pc);
  int argc; char** argv; PetscErrorCode ierr;
  ierr = PetscGetArgs(&argc, &argv); CHKERRQ(ierr);
  if (PCSetFromOptions_GenEO(c->h, argc, (const char* const*)argv)) return fail(c, "options");
  refresh_parameters(c);
  return 0;
}

}  // namespace

extern "C" {

PETSC_EXTERN PetscErrorCode createGenEOPC(PC pcPC) {       // hdr/geneo_c.h:9, geneo.cpp:2639
  if (!pcPC) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_NULL, "GenEO preconditioner is invalid");
  Ctx* c = new Ctx();
  // every handle of the reference's context starts NULL / 0 (geneo.cpp:2664-2707); the defaults of the parameters
  // come from the library (same values: geneo.cpp:2649-2662)
  c->nbDOF = c->nbDOFLoc = 0;
  c->pcMap = NULL; c->pcA = NULL; c->pcADirLoc = NULL; c->pcB = NULL; c->pcX0 = NULL; c->pcIS = NULL;
  c->dofIdxMultLoc = NULL; c->intersectLoc = NULL;
  c->pcXLoc = NULL; c->pcScatCtx = NULL; c->pcX = NULL; c->pcXOld = NULL; c->pcKSPL1Loc = NULL; c->pcDLoc = NULL;
  c->pcKSPL2 = NULL; c->pcZE2G = NULL; c->pcEEig = NULL; c->pcYEig = NULL;
  c->estimDimELoc = c->realDimELoc = c->nicolaidesLoc = 0;
  c->pcZE2GOff = NULL; c->pcEEigOff = NULL; c->pcKSPL2Off = NULL; c->pcScatCtxOff = NULL; c->pcXOff = NULL; c->pcYEigOff = NULL;
  c->tauLoc = c->gammaLoc = -1.;
  c->lvl1SetupMinvTimeLoc = 0.;
  c->lvl2SetupTauLocTimeLoc = c->lvl2SetupTauSylTimeLoc = c->lvl2SetupTauEigTimeLoc = 0.;
  c->lvl2SetupGammaLocTimeLoc = c->lvl2SetupGammaSylTimeLoc = c->lvl2SetupGammaEigTimeLoc = 0.;
  c->lvl2SetupSylTimeLoc = c->lvl2SetupEigTimeLoc = c->lvl2SetupZTimeLoc = c->lvl2SetupETimeLoc = 0.;
  c->lvl1ApplyTimeLoc = c->lvl1ApplyScatterTimeLoc = c->lvl1ApplyMinvTimeLoc = c->lvl1ApplyGatherTimeLoc = 0.;
  c->lvl1ApplyPrjFSTimeLoc = c->lvl1ApplyPrjFSZtTimeLoc = c->lvl1ApplyPrjFSEinvTimeLoc = c->lvl1ApplyPrjFSZTimeLoc = 0.;
  c->lvl2ApplyTimeLoc = c->lvl2ApplyZtTimeLoc = c->lvl2ApplyEinvTimeLoc = c->lvl2ApplyZTimeLoc = 0.;
  c->check = c->checkBin = c->checkMat = false;
  c->debug = 0; c->debugBin = c->debugMat = false;
  // One process per GPU: the rank inside the node picks the device, before the library allocates anything or RCCL
  // sees this process (ncclCommInitRank refuses two ranks on one device).  GENEO_KEEP_DEVICE=1: the launcher has bound
  // the ranks already (HIP_VISIBLE_DEVICES per rank).
  if (!getenv("GENEO_KEEP_DEVICE")) {
    MPI_Comm node;
    PetscMPIInt local = 0;
    PetscErrorCode ierr0 = MPI_Comm_split_type(PETSC_COMM_WORLD, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node); CHKERRQ(ierr0);
    ierr0 = MPI_Comm_rank(node, &local); CHKERRQ(ierr0);
    ierr0 = MPI_Comm_free(&node); CHKERRQ(ierr0);
    if (GeneoSetDevice((int)local) < 0) { delete c; SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "GenEO: no MI355X visible to this rank"); }
  }
  if (PCCreate_GenEO(&c->h)) { delete c; SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_LIB, "GenEO: cannot create the library context"); }
  refresh_parameters(c);
  // The driver asks the level-1 KSP's PC for its factor solver type (driver:946-957).  The local solves live in the
  // library; a sequential KSP with PCNONE keeps those calls valid and answers "no factor solver" (no ", L1 ..." text).
  PetscErrorCode ierr; PC l1;
  ierr = KSPCreate(PETSC_COMM_SELF, &c->pcKSPL1Loc); CHKERRQ(ierr);
  ierr = KSPGetPC(c->pcKSPL1Loc, &l1); CHKERRQ(ierr);
  ierr = PCSetType(l1, PCNONE); CHKERRQ(ierr);
  pcPC->data = (void*)static_cast<geneoContext*>(c);
  pcPC->ops->setup = setup;
  pcPC->ops->apply = apply;
  pcPC->ops->destroy = destroy;
  pcPC->ops->setfromoptions = setfromoptions;
  return 0;
}

PETSC_EXTERN PetscErrorCode PCGenEOSetup(PC pc, Mat pcADirLoc, IS dofMultiplicities, IS* dofIntersections) {   // hdr/geneo_c.h:10
  PetscErrorCode ierr; Mat P; ISLocalToGlobalMapping rmap, cmap; PetscInt n, m, N, M; const PetscInt* idx; PetscMPIInt size;
  ierr = PCGetOperators(pc, NULL, &P); CHKERRQ(ierr);
  ierr = MatGetLocalToGlobalMapping(P, &rmap, &cmap); CHKERRQ(ierr);
  if (rmap != cmap) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_WRONG, "Row and column LGMaps must match");
  ierr = MatGetSize(P, &N, &M); CHKERRQ(ierr);
  if (N != M) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_WRONG, "Matrix must be square");
  ierr = ISLocalToGlobalMappingGetSize(rmap, &n); CHKERRQ(ierr);
  ierr = ISLocalToGlobalMappingGetIndices(rmap, &idx); CHKERRQ(ierr);
  // the three vectors live as long as the PC in the reference too (it never frees them, geneo.cpp:2524-2526)
  std::vector<unsigned int>* dofs = new std::vector<unsigned int>(idx, idx + n);
  ierr = ISLocalToGlobalMappingRestoreIndices(rmap, &idx); CHKERRQ(ierr);
  ierr = ISGetLocalSize(dofMultiplicities, &m); CHKERRQ(ierr);
  if (n != m) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_WRONG, "Mismatch in dof mult size and local size");
  ierr = ISGetIndices(dofMultiplicities, &idx); CHKERRQ(ierr);
  std::vector<unsigned int>* mult = new std::vector<unsigned int>(idx, idx + n);
  ierr = ISRestoreIndices(dofMultiplicities, &idx); CHKERRQ(ierr);
  ierr = MPI_Comm_size(PETSC_COMM_WORLD, &size); CHKERRQ(ierr);
  std::vector<std::vector<unsigned int> >* inter = new std::vector<std::vector<unsigned int> >((size_t)size);
  for (int q = 0; q < size && dofIntersections; ++q) {
    ierr = ISGetLocalSize(dofIntersections[q], &m); CHKERRQ(ierr);
    ierr = ISGetIndices(dofIntersections[q], &idx); CHKERRQ(ierr);
    (*inter)[q].assign(idx, idx + m);
    ierr = ISRestoreIndices(dofIntersections[q], &idx); CHKERRQ(ierr);
  }
  return initGenEOPC(pc, (unsigned int)N, (unsigned int)n, rmap, P, pcADirLoc, NULL, NULL, dofs, mult, inter);
}

}  // extern "C"

// hdr/geneo.hpp:30-35, geneo.cpp:2591-2632: remember the inputs; the work happens in ops->setup
PetscErrorCode initGenEOPC(PC& pcPC, unsigned int const& nbDOF, unsigned int const& nbDOFLoc,
                           ISLocalToGlobalMapping const& pcMap, Mat const& pcA, Mat const& pcADirLoc, Vec const& pcB,
                           Vec const& pcX0, vector<unsigned int> const* const dofIdxDomLoc,
                           vector<unsigned int> const* const dofIdxMultLoc,
                           vector<vector<unsigned int> > const* const intersectLoc) {
  PetscErrorCode ierr;
  if (!pcPC) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_NULL, "GenEO preconditioner is invalid");
  if (!pcPC->data) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_NULL, "GenEO preconditioner without context");
  Ctx* c = ctx_of(pcPC);
  c->nbDOF = nbDOF; c->nbDOFLoc = nbDOFLoc; c->pcMap = pcMap; c->pcA = pcA;
  c->pcADirLoc = pcADirLoc; c->pcB = pcB; c->pcX0 = pcX0;
  if (pcADirLoc) { ierr = PetscObjectReference((PetscObject)pcADirLoc); CHKERRQ(ierr); }
  if (pcB) { ierr = PetscObjectReference((PetscObject)pcB); CHKERRQ(ierr); }
  if (pcX0) { ierr = PetscObjectReference((PetscObject)pcX0); CHKERRQ(ierr); }
  c->pcIS = NULL;
  if (dofIdxDomLoc) {
    std::vector<PetscInt> ids(dofIdxDomLoc->begin(), dofIdxDomLoc->end());
    ierr = ISCreateGeneral(PETSC_COMM_WORLD, (PetscInt)nbDOFLoc, ids.data(), PETSC_COPY_VALUES, &c->pcIS); CHKERRQ(ierr);
  }
  c->dofIdxMultLoc = dofIdxMultLoc;
  c->intersectLoc = intersectLoc;
  c->built = false;
  return 0;
}

// hdr/geneo.hpp:41, geneo.cpp:2274
string usageGenEO(bool const petscPrintf) {
  string msg = usageGenEO_c();
  if (petscPrintf) PetscPrintf(PETSC_COMM_WORLD, "%s", msg.c_str());
  return msg;
}
