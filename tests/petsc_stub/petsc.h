/* TEST-ONLY, declaration-only stand-in for the few PETSc / MPI names adapters/geneo_petsc_adapter.cpp and the
 * reference's hdr/geneo.hpp use.  PETSc is not installed in the build container nor on the GPU box; this header lets
 * tests/test_adapter.py COMPILE the adapter (g++ -fsyntax-only) so that its signatures, the geneoContext members it
 * touches and its calls into include/geneo_c.h are type-checked.  Nothing here is ever linked or shipped. */
#ifndef GENEO_TEST_PETSC_STUB_H
#define GENEO_TEST_PETSC_STUB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef int PetscErrorCode;
typedef int PetscInt;
typedef int PetscMPIInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef enum { PETSC_COPY_VALUES, PETSC_OWN_POINTER, PETSC_USE_POINTER } PetscCopyMode;
typedef enum { MAT_INITIAL_MATRIX, MAT_REUSE_MATRIX } MatReuse;
typedef struct _p_PetscObject* PetscObject;
typedef struct _p_PC* PC;
typedef struct _p_Mat* Mat;
typedef struct _p_Vec* Vec;
typedef struct _p_IS* IS;
typedef struct _p_KSP* KSP;
typedef struct _p_VecScatter* VecScatter;
typedef struct _p_ISLocalToGlobalMapping* ISLocalToGlobalMapping;
typedef struct _p_PetscOptionItems PetscOptionItems;
typedef const char* PCType;
typedef const char* MatType;
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#ifdef __cplusplus
#define PETSC_EXTERN extern "C"      /* as petscsys.h does for C++ translation units */
#else
#define PETSC_EXTERN extern
#endif
#define PETSC_COMM_WORLD 1
#define PETSC_COMM_SELF 2
#define PETSC_ERR_LIB 76
#define PETSC_ERR_SUP 56
#define PETSC_ERR_ARG_NULL 85
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_ARG_SIZ 60
#define MATIS "is"
#define MATAIJ "aij"
#define PCNONE "none"
#define MPI_INT 1
#define MPI_BYTE 2
#define MPI_MAX 3
#define MPI_MIN 4
#define MPI_COMM_TYPE_SHARED 1
typedef int MPI_Info;
#define MPI_INFO_NULL 0
#define MPI_IN_PLACE ((void*)1)
PetscErrorCode PetscError(MPI_Comm, int, const char*, const char*, PetscErrorCode, int, const char*, ...);
#define CHKERRQ(ierr) do { if (ierr) return (ierr); } while (0)
#define SETERRQ(c, e, s) return PetscError(c, __LINE__, "", __FILE__, e, 0, s)
#define SETERRQ1(c, e, s, a) return PetscError(c, __LINE__, "", __FILE__, e, 0, s, a)
#define SETERRQ2(c, e, s, a, b) return PetscError(c, __LINE__, "", __FILE__, e, 0, s, a, b)
PetscErrorCode PetscPrintf(MPI_Comm, const char*, ...);
PetscErrorCode PetscGetArgs(int*, char***);
PetscErrorCode PetscObjectReference(PetscObject);
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char*, PetscBool*);
PetscErrorCode PCGetOperators(PC, Mat*, Mat*);
PetscErrorCode PCSetType(PC, PCType);
PetscErrorCode KSPCreate(MPI_Comm, KSP*);
PetscErrorCode KSPGetPC(KSP, PC*);
PetscErrorCode KSPDestroy(KSP*);
PetscErrorCode MatGetLocalToGlobalMapping(Mat, ISLocalToGlobalMapping*, ISLocalToGlobalMapping*);
PetscErrorCode MatGetSize(Mat, PetscInt*, PetscInt*);
PetscErrorCode MatGetOwnershipRange(Mat, PetscInt*, PetscInt*);
PetscErrorCode MatISGetLocalMat(Mat, Mat*);
PetscErrorCode MatGetRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt*, const PetscInt**, const PetscInt**, PetscBool*);
PetscErrorCode MatSeqAIJGetArray(Mat, PetscScalar**);
PetscErrorCode MatConvert(Mat, MatType, MatReuse, Mat*);
PetscErrorCode MatCreateSubMatrices(Mat, PetscInt, const IS*, const IS*, MatReuse, Mat**);
PetscErrorCode MatDestroySubMatrices(PetscInt, Mat**);
PetscErrorCode MatDestroy(Mat*);
PetscErrorCode VecGetLocalSize(Vec, PetscInt*);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar**);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar**);
PetscErrorCode VecGetArray(Vec, PetscScalar**);
PetscErrorCode VecRestoreArray(Vec, PetscScalar**);
PetscErrorCode VecDestroy(Vec*);
PetscErrorCode ISCreateGeneral(MPI_Comm, PetscInt, const PetscInt*, PetscCopyMode, IS*);
PetscErrorCode ISGetLocalSize(IS, PetscInt*);
PetscErrorCode ISGetIndices(IS, const PetscInt**);
PetscErrorCode ISRestoreIndices(IS, const PetscInt**);
PetscErrorCode ISDestroy(IS*);
PetscErrorCode ISLocalToGlobalMappingGetSize(ISLocalToGlobalMapping, PetscInt*);
PetscErrorCode ISLocalToGlobalMappingGetIndices(ISLocalToGlobalMapping, const PetscInt**);
PetscErrorCode ISLocalToGlobalMappingRestoreIndices(ISLocalToGlobalMapping, const PetscInt**);
int MPI_Comm_size(MPI_Comm, int*);
int MPI_Comm_rank(MPI_Comm, int*);
int MPI_Bcast(void*, int, MPI_Datatype, int, MPI_Comm);
int MPI_Comm_split_type(MPI_Comm, int, int, MPI_Info, MPI_Comm*);
int MPI_Comm_free(MPI_Comm*);
int MPI_Allgather(const void*, int, MPI_Datatype, void*, int, MPI_Datatype, MPI_Comm);
int MPI_Allreduce(const void*, void*, int, MPI_Datatype, MPI_Op, MPI_Comm);
int MPI_Alltoall(const void*, int, MPI_Datatype, void*, int, MPI_Datatype, MPI_Comm);
int MPI_Alltoallv(const void*, const int*, const int*, MPI_Datatype, void*, const int*, const int*, MPI_Datatype, MPI_Comm);
#ifdef __cplusplus
}
#endif
#endif
