// geneo_petsc_adapter.cpp -- PETSc-side adapter for libgeneopc (the MI355X-native GenEO preconditioner).
//
// Drop this file into the reference tree in place of src/geneo.cpp (src/CMakeLists.txt:10 builds libgeneopc.a from it)
// and link -lgeneopc from this repository: the reference's driver (src/geneo4PETSc.cpp) then runs unchanged --
// PCRegister("geneo", createGenEOPC), PCSetFromOptions, PCGenEOSetup / initGenEOPC, KSPSetUp, KSPSolve.
//
// NOT compiled in this repository: it needs PETSc >= 3.10 (petsc.h, petsc/private/pcimpl.h), which neither the build
// container nor the GPU box has.  It only uses the C ABI of include/geneo_c.h; INTEGRATION.md walks through it.
// Build (with PETSc):  mpicxx -std=c++11 -I$PETSC_DIR/include -I<this repo>/include -c geneo_petsc_adapter.cpp
#include <petsc.h>
#include <petsc/private/pcimpl.h>          // pc->data, pc->ops (as the reference does, hdr/geneo.hpp:5)
#include <hip/hip_runtime.h>
#include <vector>
#define GENEO_HAVE_PETSC                      // PETSc's PC / PetscErrorCode stay PETSc's; our handle is GeneoPC
#include "geneo_c.h"                       // from this repository

struct Bridge { GeneoPC h = NULL; PetscInt n = 0; double *xd = NULL, *yd = NULL; };   // n: capacity of xd / yd

static PetscErrorCode csr_of(Mat seqaij, GeneoCsr* v) {         // zero-copy view of a SEQAIJ matrix
  const PetscInt *ia, *ja; PetscInt n; PetscBool ok; PetscScalar* a;
  MatGetRowIJ(seqaij, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &ok);
  MatSeqAIJGetArray(seqaij, &a);
  v->n = (int)n; v->rowptr = (const int*)ia; v->col = (const int*)ja; v->val = a;   // 32-bit PetscInt build
  return 0;
}

static PetscErrorCode setup(PC pc) {                             // ops->setup  (geneo.cpp:1672)
  Bridge* b = (Bridge*)pc->data;
  return PCSetUp_GenEO(b->h);
}
static PetscErrorCode apply(PC pc, Vec x, Vec y) {               // ops->apply  (geneo.cpp:2051)
  Bridge* b = (Bridge*)pc->data;
  const PetscScalar* xa; PetscScalar* ya; PetscInt nown;
  VecGetLocalSize(x, &nown);                                            // the rank's OWNED rows of the global Vec
  if (nown > b->n) { hipFree(b->xd); hipFree(b->yd); b->n = nown;
    hipMalloc(&b->xd, nown * sizeof(double)); hipMalloc(&b->yd, nown * sizeof(double)); }
  VecGetArrayRead(x, &xa); VecGetArray(y, &ya);
  hipMemcpy(b->xd, xa, nown * sizeof(double), hipMemcpyHostToDevice);   // PCIe copy: 2 x 8 B/DOF per apply;
  PetscErrorCode rc = PCApply_GenEO(b->h, b->xd, b->yd);                // with a HIP-enabled PETSc pass the
  hipMemcpy(ya, b->yd, nown * sizeof(double), hipMemcpyDeviceToHost);   // device arrays (VecHIPGetArray) instead
  VecRestoreArrayRead(x, &xa); VecRestoreArray(y, &ya);
  return rc;
}
static PetscErrorCode destroy(PC pc) {                           // ops->destroy (geneo.cpp:2180)
  Bridge* b = (Bridge*)pc->data;
  hipFree(b->xd); hipFree(b->yd);
  PetscErrorCode rc = PCDestroy_GenEO(&b->h);
  delete b; pc->data = NULL; return rc;
}
static PetscErrorCode setfromoptions(PetscOptionItems*, PC pc) { // ops->setfromoptions (geneo.cpp:2329)
  Bridge* b = (Bridge*)pc->data;
  int argc; char** argv; PetscGetArgs(&argc, &argv);
  return PCSetFromOptions_GenEO(b->h, argc, (const char* const*)argv);
}

extern "C" PetscErrorCode createGenEOPC(PC pc) {                 // PCRegister callback (hdr/geneo_c.h:9)
  Bridge* b = new Bridge();
  PCCreate_GenEO(&b->h);
  pc->data = b;
  pc->ops->setup = setup; pc->ops->apply = apply; pc->ops->destroy = destroy;
  pc->ops->setfromoptions = setfromoptions;
  return 0;
}

extern "C" PetscErrorCode PCGenEOSetup(PC pc, Mat pcADirLoc, IS mult, IS* inter) {   // hdr/geneo_c.h:10
  Bridge* b = (Bridge*)pc->data;
  Mat P, Aloc; ISLocalToGlobalMapping map; const PetscInt *l2g, *m; PetscInt n, N;
  PCGetOperators(pc, NULL, &P);                                   // must be MATIS (geneo.cpp:1681)
  MatGetLocalToGlobalMapping(P, &map, NULL);
  ISLocalToGlobalMappingGetIndices(map, &l2g); ISLocalToGlobalMappingGetSize(map, &n);
  MatGetSize(P, &N, NULL); MatISGetLocalMat(P, &Aloc);
  GeneoMatIS A; A.nbDOF = (int)N; A.nbDOFLoc = (int)n; A.map = (const int*)l2g; csr_of(Aloc, &A.local);
  PCSetOperators_GenEO(b->h, &A);
  GeneoCsr dir; if (pcADirLoc) csr_of(pcADirLoc, &dir);
  ISGetIndices(mult, &m);
  GeneoIS im = {(int)n, (const int*)m};
  std::vector<GeneoIS> in;                                        // GenEO-2 reads the emptiness of each list
  if (inter) { PetscMPIInt np; MPI_Comm_size(PETSC_COMM_WORLD, &np);
    for (int q = 0; q < np; ++q) { PetscInt k; ISGetLocalSize(inter[q], &k); in.push_back({(int)k, NULL}); } }
  return PCGenEOSetupViews(b->h, pcADirLoc ? &dir : NULL, im, inter ? in.data() : NULL);
}
