/*
 * geneo_c.h -- C ABI of libgeneopc, the MI355X-native GenEO preconditioner.
 *
 * Drop-in boundary for the PC-shell API of geneo4PETSc:
 *     reference hdr/geneo_c.h:9   PetscErrorCode createGenEOPC(PC);
 *     reference hdr/geneo_c.h:10  PetscErrorCode PCGenEOSetup(PC, Mat, IS, IS*);
 * and for the four callbacks createGenEOPC wires into the PETSc PC (src/geneo.cpp:2717-2720):
 *     ops->setfromoptions = setUpGenEOPCFromOptions   -> PCSetFromOptions_GenEO
 *     ops->setup          = setUpGenEOPC              -> PCSetUp_GenEO
 *     ops->apply          = applyGenEOPC              -> PCApply_GenEO
 *     ops->destroy        = destroyGenEOPC            -> PCDestroy_GenEO
 *
 * PETSc is not available on the GPU box, so the PETSc object types are replaced by plain C
 * views (pointers and sizes, no PETSc and no torch types).  With PETSc present, the adapter in
 * INTEGRATION.md pulls these views out of Mat/IS/Vec and forwards to the same entry points.
 *
 * Conventions
 *   - every function returns PetscErrorCode (int): 0 = success; on failure the message is
 *     available through PCGenEOGetError (the reference aborts through SETERRABORT,
 *     src/geneo.cpp:74; a library must not abort the host process).
 *   - host pointers are read during the call and never retained (the reference borrows
 *     pcA / pcMap / dofIdxMultLoc / intersectLoc, src/geneo.cpp:2221-2230; here they are copied).
 *   - pointers named *_dev are device (HBM) pointers of this process' GPU; vectors are FP64.
 *   - one subdomain per rank is the reference's model; this library also accepts several
 *     subdomains per rank/GPU (PCGenEOAddSubdomain), which is how one MI355X runs a whole
 *     decomposition.
 *   - not re-entrant; one context per PC; collective over the ranks given to PCGenEOSetComm.
 */
#ifndef __GENEOPC_C_ABI_H
#define __GENEOPC_C_ABI_H   /* not the reference guard (__GENEO_C_H): the PETSc-side adapter includes both headers */

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The handle.  Stand-alone it carries PETSc's names (PC, PetscErrorCode) so that host code reads like the reference's.
 * Next to the real PETSc headers (adapters/geneo_petsc_adapter.cpp) define GENEO_HAVE_PETSC before including this file:
 * PETSc's own PC / PetscErrorCode are then left alone and the handle is called GeneoPC. */
#ifdef GENEO_HAVE_PETSC
typedef struct _p_GeneoPC* GeneoPC;
#define GENEO_PC GeneoPC
#else
typedef int PetscErrorCode;
typedef struct _p_GeneoPC* PC;         /* stands for PETSc's PC */
#define GENEO_PC PC
#endif

typedef struct {                       /* stands for a SEQAIJ Mat: host CSR view */
  int n;                               /* rows = cols */
  const int* rowptr;                   /* n + 1 */
  const int* col;                      /* rowptr[n], local indices, ascending per row */
  const double* val;
} GeneoCsr;

typedef struct {                       /* stands for an IS: host index list */
  int n;
  const int* idx;
} GeneoIS;

typedef struct {                       /* stands for the MATIS operator (driver:755-757) */
  int nbDOF;                           /* global size N */
  int nbDOFLoc;                        /* local size */
  const int* map;                      /* local -> global, ascending (driver:1292-1298) */
  GeneoCsr local;                      /* local (Neumann) matrix, MatISGetLocalMat */
} GeneoMatIS;

/* ---- lifecycle ------------------------------------------------------------------------- */
/* PCCreate + PCSetType(pc,"geneo"): allocates the PC shell, then calls createGenEOPC on it. */
PetscErrorCode PCCreate_GenEO(GENEO_PC* pc);
/* src/geneo.cpp:2639-2728: (re)creates the context with the reference's default parameters. */
#ifndef GENEO_HAVE_PETSC
PetscErrorCode createGenEOPC(GENEO_PC pc);
#endif
/* The same entry under a name of its own.  A PETSc-side adapter defines the reference's createGenEOPC(PC) and
 * PCGenEOSetup(PC, Mat, IS, IS*) with PETSc's types (that is what the reference's driver links against) and forwards
 * to these two; with GENEO_HAVE_PETSC the header therefore does not declare the reference-named pair. */
PetscErrorCode PCGenEOCreateContext(GENEO_PC pc);
/* src/geneo.cpp:2180-2243 */
PetscErrorCode PCDestroy_GenEO(GENEO_PC* pc);

/* ---- options (src/geneo.cpp:2329-2514) -------------------------------------------------- */
/* Parses the -geneo_* options (same spellings, defaults, validation), plus the subset of the
 * forwarded prefixes this build understands: -els2_eps_{tol,nev,max_it,block,seed},
 * -els2_cheb_{degree,ratio}, -dls1_ksp_{rtol,max_it}, -ksp_{type,rtol,atol,divtol,max_it},
 * -ksp_gmres_restart.  Unknown options are ignored, like PETSc with -options_left no. */
PetscErrorCode PCSetFromOptions_GenEO(GENEO_PC pc, int argc, const char* const* argv);
PetscErrorCode PCGenEOSetOption(GENEO_PC pc, const char* key, const char* value);
/* buildGenEOName, src/geneo.cpp:2245-2268 ("geneo1ASM", "geneo1HASM", ...) */
const char* PCGenEOGetName(GENEO_PC pc);
const char* PCGenEOGetError(GENEO_PC pc);
/* the parsed options as "key=value;..." (what the driver reads from the public geneoContext fields,
 * src/geneo4PETSc.cpp:928-989) */
const char* PCGenEOGetOptionsString(GENEO_PC pc);
/* usageGenEO, src/geneo.cpp:2274-2327 */
const char* usageGenEO_c(void);

/* ---- inputs ----------------------------------------------------------------------------- */
/* KSPSetOperators(ksp, A, A) with A of type MATIS (required: src/geneo.cpp:1681). */
PetscErrorCode PCSetOperators_GenEO(GENEO_PC pc, const GeneoMatIS* A);
/* hdr/geneo_c.h:10, src/geneo.cpp:2518-2572: one subdomain per rank, operator taken from
 * PCSetOperators_GenEO; pcADirLoc may be NULL (built from A); dofIntersections may be NULL
 * (only its emptiness pattern is used, and only by GenEO-2). */
#ifndef GENEO_HAVE_PETSC
PetscErrorCode PCGenEOSetup(GENEO_PC pc, const GeneoCsr* pcADirLoc, GeneoIS dofMultiplicities,
                            const GeneoIS* dofIntersections);
#endif
PetscErrorCode PCGenEOSetupViews(GENEO_PC pc, const GeneoCsr* pcADirLoc, GeneoIS dofMultiplicities,
                                 const GeneoIS* dofIntersections);   /* = PCGenEOSetup, see PCGenEOCreateContext */
/* C form of initGenEOPC (hdr/geneo.hpp:30-35, src/geneo.cpp:2591-2632). b_dev / x0_dev may be NULL. */
PetscErrorCode initGenEOPC_c(GENEO_PC pc, unsigned int nbDOF, unsigned int nbDOFLoc, const int* map,
                             const GeneoCsr* A_local, const GeneoCsr* ADirLoc, const double* b_dev,
                             double* x0_dev, const unsigned int* dofIdxMultLoc);

/* Several subdomains on one rank / GPU (extension; gid = the MPI rank the reference would use). */
PetscErrorCode PCGenEOSetSizes(GENEO_PC pc, int nbDOF, int nbSubdomainsGlobal);
PetscErrorCode PCGenEOAddSubdomain(GENEO_PC pc, int gid, int nbDOFLoc, const int* map, const int* multiplicity,
                                   const GeneoCsr* A_local, const GeneoCsr* ADirLoc);

/* intersectLoc of initGenEOPC (hdr/geneo.hpp:34): nonempty[q] != 0 iff subdomain gid shares DOFs with
 * subdomain q.  Only GenEO-2's gamma_loc reads it (src/geneo.cpp:1139-1148); on one rank it is derived
 * from the maps when absent.  gid < 0 addresses the subdomain added last. */
PetscErrorCode PCGenEOSetIntersect(GENEO_PC pc, int gid, int nbSubdomainsGlobal, const int* nonempty);

/* ---- multi-rank plumbing (one process per GPU; the transport is supplied by the host) ----- */
typedef int (*GeneoExchangeFn)(void* user, int reverse);  /* 0 = forward (owner -> halo), 1 = reverse */
typedef int (*GeneoAllreduceFn)(void* user, int n);       /* in-place sum of red_dev[0..n) over ranks */
/* owned_gid: ascending global ids owned by this rank.  halo_gid: ids this rank reads but does not
 * own, grouped by owner rank (recv_counts[q] ids from rank q).  send_idx: owned-local indices this
 * rank sends, grouped by destination (send_counts[q]).  The callbacks move send_dev -> recv_dev
 * with these counts (forward) or with the two count arrays swapped (reverse), on the stream given
 * to GeneoSetStream.  Buffers are device memory owned by the caller (>= max(sum send, sum recv)). */
PetscErrorCode PCGenEOSetComm(GENEO_PC pc, int rank, int size, int n_owned, const int* owned_gid, int n_halo,
                              const int* halo_gid, const int* recv_counts, const int* send_counts,
                              const int* send_idx, GeneoExchangeFn exchange, GeneoAllreduceFn allreduce,
                              void* user, double* send_dev, double* recv_dev, double* red_dev, int red_capacity);
/* Optional: send_dev / recv_dev hold max_width x the single-vector counts (entry-major, the vectors of one entry
 * contiguous) and the exchange callback honours the width passed in the upper bits of its flag
 * (flag = reverse | width << 1; width 0 means 1).  Lets the coarse operator E be assembled 32 columns per
 * exchange instead of one.  Default width 1. */
PetscErrorCode PCGenEOSetCommWidth(GENEO_PC pc, int max_width);

/* C++ transport over RCCL / xGMI built into the library (csrc/comm_rccl.cpp): the counterpart of the reference's
 * VecScatter (src/geneo.cpp:156,:1850,:1881) and MPI reductions (:1474) for one process per GPU.  Halo exchanges are
 * ncclSend / ncclRecv groups with the neighbours, reductions ncclAllReduce, all on the stream given to GeneoSetStream.
 * Bootstrap: rank 0 calls GeneoRcclUniqueId and the HOST broadcasts the 128 bytes (MPI_Bcast, torch.distributed, ...);
 * every rank then calls GeneoRcclCreate (collective) and attaches its PCs with PCGenEOSetCommRccl, which allocates the
 * device buffers (max_width vectors per exchange, cf. PCGenEOSetCommWidth) and installs the two callbacks of
 * PCGenEOSetComm.  RCCL is resolved with dlopen: no link-time dependency. */
typedef struct _p_GeneoRccl* GeneoRccl;
PetscErrorCode GeneoRcclUniqueId(char* id128);
PetscErrorCode GeneoRcclCreate(const char* id128, int rank, int size, GeneoRccl* comm);
PetscErrorCode PCGenEOSetCommRccl(GENEO_PC pc, GeneoRccl comm, int n_owned, const int* owned_gid, int n_halo,
                                  const int* halo_gid, const int* recv_counts, const int* send_counts,
                                  const int* send_idx, int max_width);
PetscErrorCode GeneoRcclDestroy(GeneoRccl* comm);
const char* GeneoRcclGetError(void);
/* bring-up hooks: the device buffers of the which-th attached plan and direct calls of its two callbacks */
PetscErrorCode GeneoRcclPlanBuffers(GeneoRccl comm, int which, double** send_dev, double** recv_dev, double** red_dev);
PetscErrorCode GeneoRcclPlanExchange(GeneoRccl comm, int which, int flag);
PetscErrorCode GeneoRcclPlanAllreduce(GeneoRccl comm, int which, int n);

/* ---- PC operations (the PETSc ops table, src/geneo.cpp:2717-2720) ------------------------- */
PetscErrorCode PCSetUp_GenEO(GENEO_PC pc);                                   /* setUpGenEOPC :1672 */
PetscErrorCode PCApply_GenEO(GENEO_PC pc, const double* x_dev, double* y_dev); /* applyGenEOPC :2051 */
PetscErrorCode PCGenEOApplyQ(GENEO_PC pc, const double* x_dev, double* y_dev); /* applyQ :1435 */
PetscErrorCode MatMult_GenEO(GENEO_PC pc, const double* x_dev, double* y_dev); /* MatMult on the MATIS A */
/* initial guess written by setup (src/geneo.cpp:1601-1607): Q b for the efficient hybrid, else 0 */
PetscErrorCode PCGenEOGetX0(GENEO_PC pc, double* x0_dev);
PetscErrorCode PCGenEOSetRHS(GENEO_PC pc, const double* b_dev);

/* ---- Krylov driver (counterpart of KSPSolve at src/geneo4PETSc.cpp:1240; PETSc's own KSP
 *      drives PCApply_GenEO instead when PETSc is present) ----------------------------------- */
PetscErrorCode KSPSolve_GenEO(GENEO_PC pc, const double* b_dev, double* x_dev, int* its, double* rnorm, int* reason);
int PCGenEOGetResidualHistory(GENEO_PC pc, double* hist, int cap);

/* ---- public counters / timers of geneoContext (hdr/geneo.hpp:96-123) ----------------------- */
typedef struct {
  int estimDimELoc, realDimELoc, nicolaidesLoc, dimE;
  int eig_iterations, eig_spmm;
  long long dls1_iterations, dls1_solves, spmv_calls;
  double lvl1SetupMinvTimeLoc, lvl2SetupEigTimeLoc, lvl2SetupZTimeLoc, lvl2SetupETimeLoc;
  double lvl1ApplyTimeLoc, lvl1ApplyScatterTimeLoc, lvl1ApplyMinvTimeLoc, lvl1ApplyGatherTimeLoc;
  double lvl1ApplyPrjFSTimeLoc, lvl2ApplyTimeLoc, lvl2ApplyZtTimeLoc, lvl2ApplyEinvTimeLoc, lvl2ApplyZTimeLoc;
  double setupTime, solveTime;
  int amg_levels;                       /* levels of the inner AMG hierarchy (0 = not used) */
  double amg_operator_complexity, amgSetupTime;
  int nullPivotsLoc;                    /* null pivots detected and fixed in the local factorisations of this set-up (singular
                                           subdomain matrices; the reference's MUMPS settings of tuneSolver, geneo.cpp:76-92) */
  int eigGroups;                        /* consecutive subdomain groups the local eigensolves of this set-up ran in (1: all at once;
                                           > 1: memory-bounded set-up, -geneo_eig_group_rows / -geneo_eig_mem_gb) */
  int eigCoarseIterations;              /* LOBPCG iterations spent on the multigrid level-1 pencil whose Ritz vectors start the fine
                                           eigensolve (-geneo_eig_coarse_start; 0: the fine eigensolve started from its random block) */
} GeneoInfo;
PetscErrorCode PCGenEOGetInfo(GENEO_PC pc, GeneoInfo* info);
/* eigenvalues kept in Z for local subdomain s (returns the count; copies min(count, cap)) */
int PCGenEOGetEigenvalues(GENEO_PC pc, int local_sub, double* vals, int cap);
int PCGenEOGetCandidates(GENEO_PC pc, int local_sub, double* vals, int cap);
/* coarse operator E (dimE x dimE row-major); returns dimE */
int PCGenEOGetE(GENEO_PC pc, double* e, int cap);
int PCGenEOGetLocalDims(GENEO_PC pc, int* ksub_global, int cap);
/* tau_loc / gamma_loc per local subdomain (getLocalGenEOTau / Gamma, src/geneo.cpp:1097-1232); returns the count */
int PCGenEOGetLocalParams(GENEO_PC pc, double* tau_loc, double* gamma_loc, int cap);

/* ---- input plugins ------------------------------------------------------------------------------
 * The reference driver loads its operators from `--inpLibA lib.so#args` plugins exporting the C++ function
 * `getInput` (src/geneo4PETSc.cpp:75-96; tst/laplacian, tst/heat, tst/graph).  GeneoGetLibInput loads such a
 * plugin unchanged ('#' in the arguments stands for a blank, as on the reference CLI) and returns the element
 * list flattened: element e has nodes elemIdx[elemPtr[e] .. elemPtr[e+1]) and a row-major k x k matrix, the
 * matrices stored back to back in elemMat.  Free with GeneoFreeInput. */
typedef struct {
  unsigned int nbElem, nbNode;
  unsigned int* elemPtr;   /* nbElem + 1 */
  unsigned int* elemIdx;   /* nIdx */
  double* elemMat;         /* nMat */
  size_t nIdx, nMat;
} GeneoInput;
PetscErrorCode GeneoGetLibInput(const char* inpLibA, const char* inpLibArg, GeneoInput* out);
void GeneoFreeInput(GeneoInput* in);

/* ---- host-side decomposition and structured generators (csrc/decompose.cpp) ---------------------------------------
 * The reference DRIVER's decompose + addOverlapLayers + buildDomain + fillALoc (src/geneo4PETSc.cpp:196-379, :447-494,
 * :643-715): from the element lists and a k-way partition (elemPart when dual, nodePart when nodal) to each domain's
 * ascending global node list, multiplicities, MATIS local matrix A_Neu (element matrices weighted 1 / elemMult) and
 * A_Dir = R A R^T, plus the lists of local indices shared with every other domain (intersectLoc, hdr/geneo.hpp:34).
 * nodes: nbElem x W node ids (-1: unused slot), mats: nbElem x W*W; both must outlive the handle.  GeneoDomain arrays
 * are malloc'ed by the library: release with GeneoFreeDomain.  Host code only. */
typedef struct _p_GeneoDecomp* GeneoDecomp;
typedef struct {
  int n;                                /* local DOFs */
  int *l2g, *mult;                      /* n each */
  int *neu_rowptr, *neu_col; double* neu_val;
  int *dir_rowptr, *dir_col; double* dir_val;     /* NULL without withDirichlet */
  int *inter_ptr, *inter_idx;           /* nbPart + 1 offsets into the shared local indices, per other part */
} GeneoDomain;
PetscErrorCode GeneoDecompCreate(int nbNode, int nbElem, int W, const int* nodes, const double* mats, int nbPart,
                                 const int* elemPart, const int* nodePart, int dual, int addOverlap, GeneoDecomp* out);
PetscErrorCode GeneoDecompDomain(GeneoDecomp d, int p, int withDirichlet, GeneoDomain* out);
void GeneoFreeDomain(GeneoDomain* dom);
void GeneoDecompDestroy(GeneoDecomp* d);
/* tst/laplacian (laplacian.cpp:57-188) / tst/heat (heat.cpp:64-261) generator on an n^dim grid (interp: 0 none, 1 quad,
 * 2 lin, 3 minmax), optionally restricted to the elements inside the node-index box [wlo, whi) (NULL: whole grid).
 * nodes (nbElem x 2, -1 in the second slot of a Dirichlet element) and mats (nbElem x 4) are malloc'ed: GeneoFreeMesh. */
PetscErrorCode GeneoGridMesh(int n, int dim, double inpEps, double kappaMax, int interp, int heat, double lbd, double dt,
                             const int* wlo, const int* whi, int* nbNode, int* nbElem, int** nodes, double** mats);
void GeneoFreeMesh(int* nodes, double* mats);

/* ---- host k-way mesh partitioner (csrc/partition.cpp): what the reference's driver asks of Metis 5.1.0 --------------
 * METIS_PartMeshDual(&ne, &nn, eptr, eind, NULL, NULL, &ncommon = 1, &nparts, NULL, options, &objval, epart, npart) and
 * METIS_PartMeshNodal(&ne, &nn, eptr, eind, NULL, NULL, &nparts, NULL, options, &objval, epart, npart) with
 * PTYPE_KWAY / OBJTYPE_CUT (src/geneo4PETSc.cpp:381-445).  Elements are node lists (eptr[ne + 1], eind[eptr[ne]]);
 * epart[ne] / npart[nn] receive the parts; objval the edge cut of the partitioned graph.  Multilevel recursive bisection
 * (heavy-edge matching, graph growing + spectral initial cuts, FM refinement, multilevel spectral candidate); sizes exact
 * to one vertex per bisection; deterministic.  Host code only (no device is touched).  0 = ok. */
int GeneoPartMeshDual(int ne, int nn, const int* eptr, const int* eind, int nparts, int* objval, int* epart, int* npart);
int GeneoPartMeshNodal(int ne, int nn, const int* eptr, const int* eind, int nparts, int* objval, int* epart, int* npart);
/* METIS_PartGraphKway on a CSR graph (symmetric adjacency, no self loops, unit weights) */
int GeneoPartGraphKway(int n, const int* xadj, const int* adjncy, int nparts, int* objval, int* part);

/* ---- the CLI driver (csrc/driver_main.cpp): counterpart of the reference executable's main() (src/geneo4PETSc.cpp:1569)
 * with its flags (--inpFileA / --inpLibA / --inpFileB / --inpEps / --metisDual / --metisNodal / --addOverlap / --verbose /
 * --timing / --shortRes / --cmdLine; --np N for `mpirun -n N`, --parts / --partFile for a given partition; everything else
 * goes to the PC) and its INFO: / TIME: output lines (driver:898-1231) on stdout.  argv WITHOUT the program name.
 * The executable geneo4petsc_amd/geneo_driver is a main() over it.  0 = ok. */
int GeneoDriverMain(int argc, const char* const* argv);

/* ---- device helpers for hosts without a HIP runtime of their own ---------------------------- */
const char* GeneoBackendName(void);              /* "hip-gfx950" in the product library */
PetscErrorCode GeneoSetStream(void* hip_stream); /* all launches / copies go to this stream */
/* One process per GPU: bind this process to GPU (local_rank mod GeneoDeviceCount()) BEFORE the first allocation, set-up
 * or RCCL call (ncclCommInitRank refuses two ranks on one device).  Returns the device ordinal, -1 on failure.  The
 * reference has no counterpart (CPU only); under mpirun pass the rank inside the node (MPI_Comm_split_type(SHARED)),
 * as adapters/geneo_petsc_adapter.cpp does -- or give every rank its own HIP_VISIBLE_DEVICES and skip the call. */
int GeneoDeviceCount(void);
int GeneoSetDevice(int local_rank);
/* The HIP current device belongs to the host THREAD: the library binds every thread it starts (side-stream set-up, upload
 * helpers) to the device of the thread that configured it (GeneoSetDevice, GeneoSetStream or the first allocation).
 * GeneoCurrentDevice: that device (-1 before the first call).  GeneoThreadDeviceCheck (test hook): starts a thread the
 * way the library does and returns the device it ends up on -- equal to GeneoCurrentDevice() or the binding is broken. */
int GeneoCurrentDevice(void);
int GeneoThreadDeviceCheck(void);
/* hipFree every device block the library's caching allocator is holding (it keeps freed blocks for the next set-up, up
 * to GENEO_ALLOC_CACHE_GB, default 96): call it when another allocator of the process needs the memory. */
void GeneoAllocCacheRelease(void);
void* GeneoDeviceAlloc(size_t bytes);
void GeneoDeviceFree(void* p);
PetscErrorCode GeneoH2D(void* dst_dev, const void* src, size_t bytes);
PetscErrorCode GeneoD2H(void* dst, const void* src_dev, size_t bytes);
PetscErrorCode GeneoDeviceSync(void);
int GeneoSelfTestMFMA(void);                     /* 0 = f64 MFMA lane maps as assumed */
/* y = a x + b y on device vectors (calibration kernel of the HBM-traffic PMC passes) */
PetscErrorCode GeneoTestAxpby(double* y_dev, const double* x_dev, double a, double b, int n);
PetscErrorCode GeneoSetSpmvKind(int kind);       /* 0: LDS row-block SpMV kernel, 1: 64-row sliced kernel (default) */
const char* GeneoSpmvKernelName(void);
PetscErrorCode GeneoSetMFMA(int enable);         /* 0: run the plain-FMA twins of the MFMA kernels (validation) */
/* validation: choose between two device forms of the same arithmetic -- "spgemm_fill_scan" (1: owner-computes numeric pass
 * of the sparse products instead of the hash accumulators), "spgemm_small_rows" (0: rows of at most 64 products go through
 * the hash table as well), "gram_flat" (0: k_gram_mfma instead of k_gram_flat); both forms of each give bit-identical
 * results (tests/test_gpu_kernels.py).  Returns 1 for an unknown name. */
PetscErrorCode GeneoSetKernelVariant(const char* name, int value);

/* ---- stand-alone kernels (parity tests and the roofline leg of bench.py) --------------------- */
typedef struct _p_GeneoSpmv* GeneoSpmv;
PetscErrorCode GeneoSpmvCreate(const GeneoCsr* a, GeneoSpmv* h);
PetscErrorCode GeneoSpmvApply(GeneoSpmv h, const double* x_dev, double* y_dev);
/* average kernel time of `reps` back-to-back launches, HIP events on the library stream */
PetscErrorCode GeneoSpmvTime(GeneoSpmv h, const double* x_dev, double* y_dev, int reps, double* ms_avg);
PetscErrorCode GeneoSpmvDestroy(GeneoSpmv* h);
/* in-situ timing of every `every`-th CSR SpMV launch the library issues (solve loops included):
 * HIP events on the launch stream; stop returns the summed kernel ms and algorithmic bytes. */
PetscErrorCode GeneoSpmvProfileStart(int every, double min_bytes);  /* launches moving < min_bytes are skipped */
PetscErrorCode GeneoSpmvProfileStop(double* ms_sum, double* bytes_sum, long long* nsampled, long long* nlaunch);
/* the same for every hot kernel class at once: 0 fine-level CSR SpMV, 1 fine-level SpMM (16 / 32 / 64 columns),
 * 2 MFMA Gram (S^T T), 3 MFMA block update (S C).  Start ... the library's work ... Stop, then Get per class: summed
 * kernel ms, algorithmic bytes and flops of the SAMPLED launches, their number and the number of launches seen. */
PetscErrorCode GeneoKernelProfileStart(int every, double spmv_min_bytes);
/* Device memory of the library's own blocks, in bytes (any pointer may be NULL): handed out now, their high-water mark,
 * the high-water mark of handed out + parked in the caching allocator (the library's footprint on the card), parked now,
 * and hipMemGetInfo's free / total.  reset_peaks != 0 restarts both high-water marks at the current state. */
PetscErrorCode GeneoDeviceMemInfo(double* live, double* live_peak, double* footprint_peak, double* cached, double* dev_free,
                                  double* dev_total, int reset_peaks);
PetscErrorCode GeneoKernelProfileStop(void);
PetscErrorCode GeneoKernelProfileGet(int kernel_class, double* ms_sum, double* bytes_sum, double* flops_sum,
                                     long long* nsampled, long long* nlaunch);
/* Y = post.*(A (pre.*X)), row-major n x m blocks */
PetscErrorCode GeneoSpmmApply(GeneoSpmv h, const double* X_dev, double* Y_dev, int m, const double* pre_dev,
                              const double* post_dev);
/* the same on strided blocks (leading dimensions ldx / ldy >= m), timed: average kernel time of `reps` back-to-back
 * launches, HIP events on the library stream (reps <= 0: one untimed launch, ms_avg untouched) */
PetscErrorCode GeneoSpmmTime(GeneoSpmv h, const double* X_dev, int ldx, double* Y_dev, int ldy, int m,
                             const double* pre_dev, const double* post_dev, int reps, double* ms_avg);
/* multigrid epilogues fused into the SpMV (m = 1) / SpMM launch, row-major n x m blocks:
 *   epi 1: Y = B - A X      2: Y = Z + A X      3: Y = X + w dinv.*(B - A X)      4: Z = w dinv.*B, Y = B - A Z
 *   epi 5: Y = w dinv.*(Z + B) + A X (prolongation + correction + post-smoothing sweep in one product) */
PetscErrorCode GeneoSpmmFused(GeneoSpmv h, int epi, const double* X_dev, double* Y_dev, int m, const double* B_dev,
                              double* Z_dev, const double* dinv_dev, double w);
/* the single-vector launches (m = 1; epi 0: Y = A X) reading the matrix's single-precision companion -- float values,
 * 16-bit column offsets per 64-row slice (32-bit columns when a slice spans more than 65535), FP64 arithmetic: what the V-cycle of the local solves streams
 * (-dls1_amg_precision single).  Error when the matrix has no such companion. */
/* Y1 = B X, Y2 = A X in ONE pass over X with B laid out on A's sliced pattern (LOBPCG's A W / B W pass); 2 = pattern(B)
 * not contained in pattern(A) or A not on the sliced path */
PetscErrorCode GeneoSpmmDualTest(GeneoSpmv a, GeneoSpmv b, const double* X_dev, int ldx, double* Y1_dev, double* Y2_dev,
                                 int ldy, int m);
/* R = mask .* (A X - B X diag(lam)) per subdomain, both products in one pass over X and neither written (the residual
 * block of LOBPCG's lean iteration); suboff: nsub + 1 first rows, lam / mask: nsub x m HOST arrays; X_dev, R_dev device */
PetscErrorCode GeneoSpmmDualResidualTest(GeneoSpmv a, GeneoSpmv b, const double* X_dev, int ldx, double* R_dev, int ldr,
                                         int m, int nsub, const int* suboff, const double* lam, const double* mask);
PetscErrorCode GeneoSpmvFusedSingle(GeneoSpmv h, int epi, const double* X_dev, double* Y_dev, const double* B_dev,
                                    double* Z_dev, const double* dinv_dev, double w);
/* device sparse products of the multigrid set-up (test hook): op 0: C = A B, op 1: C = A^T; returns nnz(C), -1 when a
 * row exceeds the kernels' per-row capacity (callers fall back to the host product), -2 on error */
long long GeneoTestSparseProduct(int op, const GeneoCsr* A, const GeneoCsr* B, int ncols, int* rowptr_out, int* col_out,
                                 double* val_out, long long cap);
/* per-subdomain tall-skinny kernels on host data (suboff: nsub+1 row offsets):
 *   kind 0: G[s] = S_s^T T_s (p x q)     kind 1: Y_s = S_s C_s (C: nsub x p x q)
 *   kind 2: kind 0 through the two-left-block entry (columns [0, p/2) and [p/2, p) of S passed as separate views)
 * GeneoSetMFMA(0) selects the plain-FMA twin.  reps > 0 also times it (HIP events). */
PetscErrorCode GeneoBlockKernel(int kind, int nsub, const int* suboff, const double* S, int p, const double* T_or_C,
                                int q, double* out, int reps, double* ms_avg);

/* Per-subdomain reductions (dot products of the batched PCG, Z^T x, Gram partials) switch to their cooperative forms --
 * one workgroup per subdomain, once per launch -- above this many 1024-row chunks per subdomain (default 1024, also
 * GENEO_PAR_REDUCE_MIN): the one-subdomain-per-GPU layout of the benchmark's configuration.  Returns the previous value;
 * set it before creating the PC it should govern. */
int GeneoSetParReduceMin(int chunks);
/* fused Rayleigh-Ritz update of one LOBPCG iteration, m = 32 (test hook; host arrays): S, AS, BS n x 96 row-major
 * [X | P | W]; C nsub x 96 x 64; keep / lam / mask nsub x 32.  Out: columns 0..63 of T, AT, BT (n x 96) = [X' P'] of each
 * operand, R (n x 32) = mask .* (A X' - B X' diag(lam)).  AS == NULL: the basis-only form (T from S, C, keep; the other
 * arguments are not touched). */
PetscErrorCode GeneoTestLobpcgUpdate(int nsub, const int* suboff, const double* S, const double* AS, const double* BS,
                                     const double* C, const double* keep, const double* lam, const double* mask,
                                     double* T, double* AT, double* BT, double* R);

#ifdef __cplusplus
}
#endif
#endif
