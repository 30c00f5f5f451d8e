/* TEST-ONLY stand-in for petsc/private/pcimpl.h: the two members of the PC object the reference (hdr/geneo.hpp:5,
 * src/geneo.cpp:2717-2720) and the adapter touch -- the ops table and the data pointer. */
#ifndef GENEO_TEST_PCIMPL_STUB_H
#define GENEO_TEST_PCIMPL_STUB_H
#include <petsc.h>
struct _PCOps {
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*apply)(PC, Vec, Vec);
  PetscErrorCode (*destroy)(PC);
  PetscErrorCode (*setfromoptions)(PetscOptionItems*, PC);
};
struct _p_PC {
  struct _PCOps ops[1];
  void* data;
};
#endif
