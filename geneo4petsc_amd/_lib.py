"""ctypes binding of libgeneopc.so (C ABI declared in include/geneo_c.h).

The product path is HIP only: ``load()`` loads the in-tree ``libgeneopc.so`` and raises if it is
missing or if it is not the gfx950 build.  There is no CPU fallback in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgeneopc.so")

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


class GeneoCsr(C.Structure):
    _fields_ = [("n", C.c_int), ("rowptr", c_int_p), ("col", c_int_p), ("val", c_dbl_p)]


class GeneoIS(C.Structure):
    _fields_ = [("n", C.c_int), ("idx", c_int_p)]


class GeneoMatIS(C.Structure):
    _fields_ = [("nbDOF", C.c_int), ("nbDOFLoc", C.c_int), ("map", c_int_p), ("local", GeneoCsr)]


class GeneoInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("estimDimELoc", "realDimELoc", "nicolaidesLoc", "dimE",
                                       "eig_iterations", "eig_spmm")] + \
               [(n, C.c_longlong) for n in ("dls1_iterations", "dls1_solves", "spmv_calls")] + \
               [(n, C.c_double) for n in ("lvl1SetupMinvTimeLoc", "lvl2SetupEigTimeLoc", "lvl2SetupZTimeLoc",
                                          "lvl2SetupETimeLoc", "lvl1ApplyTimeLoc", "lvl1ApplyScatterTimeLoc",
                                          "lvl1ApplyMinvTimeLoc", "lvl1ApplyGatherTimeLoc",
                                          "lvl1ApplyPrjFSTimeLoc", "lvl2ApplyTimeLoc", "lvl2ApplyZtTimeLoc",
                                          "lvl2ApplyEinvTimeLoc", "lvl2ApplyZTimeLoc", "setupTime", "solveTime")] + \
               [("amg_levels", C.c_int), ("amg_operator_complexity", C.c_double), ("amgSetupTime", C.c_double),
                ("nullPivotsLoc", C.c_int), ("eigGroups", C.c_int),
                ("eigCoarseIterations", C.c_int)]


class GeneoDomain(C.Structure):
    """include/geneo_c.h GeneoDomain: one domain of GeneoDecompDomain (arrays owned by the library: GeneoFreeDomain)"""
    _fields_ = [("n", C.c_int), ("l2g", c_int_p), ("mult", c_int_p),
                ("neu_rowptr", c_int_p), ("neu_col", c_int_p), ("neu_val", c_dbl_p),
                ("dir_rowptr", c_int_p), ("dir_col", c_int_p), ("dir_val", c_dbl_p),
                ("inter_ptr", c_int_p), ("inter_idx", c_int_p)]


class GeneoInput(C.Structure):
    _fields_ = [("nbElem", C.c_uint), ("nbNode", C.c_uint), ("elemPtr", C.POINTER(C.c_uint)),
                ("elemIdx", C.POINTER(C.c_uint)), ("elemMat", c_dbl_p), ("nIdx", C.c_size_t), ("nMat", C.c_size_t)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)

# every symbol include/geneo_c.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "PCCreate_GenEO": (C.c_int, [C.POINTER(C.c_void_p)]),
    "createGenEOPC": (C.c_int, [C.c_void_p]),
    "PCGenEOCreateContext": (C.c_int, [C.c_void_p]),
    "PCDestroy_GenEO": (C.c_int, [C.POINTER(C.c_void_p)]),
    "PCSetFromOptions_GenEO": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]),
    "PCGenEOSetOption": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "PCGenEOGetName": (C.c_char_p, [C.c_void_p]),
    "PCGenEOGetError": (C.c_char_p, [C.c_void_p]),
    "PCGenEOGetOptionsString": (C.c_char_p, [C.c_void_p]),
    "usageGenEO_c": (C.c_char_p, []),
    "PCSetOperators_GenEO": (C.c_int, [C.c_void_p, C.POINTER(GeneoMatIS)]),
    "PCGenEOSetup": (C.c_int, [C.c_void_p, C.POINTER(GeneoCsr), GeneoIS, C.POINTER(GeneoIS)]),
    "PCGenEOSetupViews": (C.c_int, [C.c_void_p, C.POINTER(GeneoCsr), GeneoIS, C.POINTER(GeneoIS)]),
    "initGenEOPC_c": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint, c_int_p, C.POINTER(GeneoCsr),
                                C.POINTER(GeneoCsr), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]),
    "PCGenEOSetSizes": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "PCGenEOAddSubdomain": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int_p, c_int_p, C.POINTER(GeneoCsr),
                                      C.POINTER(GeneoCsr)]),
    "PCGenEOSetComm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int_p, C.c_int, c_int_p, c_int_p,
                                 c_int_p, c_int_p, EXCHANGE_FN, ALLREDUCE_FN, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int]),
    "GeneoRcclUniqueId": (C.c_int, [C.c_char_p]),
    "GeneoRcclCreate": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "PCGenEOSetCommRccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_int_p, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p,
                                     C.c_int]),
    "GeneoRcclDestroy": (C.c_int, [C.POINTER(C.c_void_p)]),
    "GeneoRcclGetError": (C.c_char_p, []),
    "GeneoRcclPlanBuffers": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "GeneoRcclPlanExchange": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "GeneoRcclPlanAllreduce": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "PCSetUp_GenEO": (C.c_int, [C.c_void_p]),
    "PCApply_GenEO": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "PCGenEOApplyQ": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "MatMult_GenEO": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "PCGenEOGetX0": (C.c_int, [C.c_void_p, C.c_void_p]),
    "PCGenEOSetRHS": (C.c_int, [C.c_void_p, C.c_void_p]),
    "KSPSolve_GenEO": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_dbl_p, c_int_p]),
    "PCGenEOGetResidualHistory": (C.c_int, [C.c_void_p, c_dbl_p, C.c_int]),
    "PCGenEOGetInfo": (C.c_int, [C.c_void_p, C.POINTER(GeneoInfo)]),
    "PCGenEOGetEigenvalues": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p, C.c_int]),
    "PCGenEOGetCandidates": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p, C.c_int]),
    "PCGenEOGetE": (C.c_int, [C.c_void_p, c_dbl_p, C.c_int]),
    "PCGenEOGetLocalDims": (C.c_int, [C.c_void_p, c_int_p, C.c_int]),
    "PCGenEOGetLocalParams": (C.c_int, [C.c_void_p, c_dbl_p, c_dbl_p, C.c_int]),
    "PCGenEOSetCommWidth": (C.c_int, [C.c_void_p, C.c_int]),
    "PCGenEOSetIntersect": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int_p]),
    "GeneoGetLibInput": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(GeneoInput)]),
    "GeneoFreeInput": (None, [C.POINTER(GeneoInput)]),
    "GeneoTestSparseProduct": (C.c_longlong, [C.c_int, C.POINTER(GeneoCsr), C.POINTER(GeneoCsr), C.c_int, c_int_p, c_int_p, c_dbl_p,
                               C.c_longlong]),
    "GeneoBackendName": (C.c_char_p, []),
    "GeneoSetStream": (C.c_int, [C.c_void_p]),
    "GeneoDecompCreate": (C.c_int, [C.c_int, C.c_int, C.c_int, c_int_p, c_dbl_p, C.c_int, c_int_p, c_int_p, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p)]),
    "GeneoDecompDomain": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(GeneoDomain)]),
    "GeneoFreeDomain": (None, [C.POINTER(GeneoDomain)]),
    "GeneoDecompDestroy": (None, [C.POINTER(C.c_void_p)]),
    "GeneoGridMesh": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, c_int_p, c_int_p,
                                c_int_p, c_int_p, C.POINTER(c_int_p), C.POINTER(c_dbl_p)]),
    "GeneoFreeMesh": (None, [c_int_p, c_dbl_p]),
    "GeneoPartMeshDual": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    "GeneoPartMeshNodal": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    "GeneoPartGraphKway": (C.c_int, [C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p]),
    "GeneoDeviceCount": (C.c_int, []),
    "GeneoSetDevice": (C.c_int, [C.c_int]),
    "GeneoCurrentDevice": (C.c_int, []),
    "GeneoThreadDeviceCheck": (C.c_int, []),
    "GeneoAllocCacheRelease": (None, []),
    "GeneoDeviceAlloc": (C.c_void_p, [C.c_size_t]),
    "GeneoDeviceFree": (None, [C.c_void_p]),
    "GeneoH2D": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "GeneoD2H": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "GeneoDeviceSync": (C.c_int, []),
    "GeneoSelfTestMFMA": (C.c_int, []),
    "GeneoSetMFMA": (C.c_int, [C.c_int]),
    "GeneoSetKernelVariant": (C.c_int, [C.c_char_p, C.c_int]),
    "GeneoTestAxpby": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int]),
    "GeneoSetSpmvKind": (C.c_int, [C.c_int]),
    "GeneoSpmvKernelName": (C.c_char_p, []),
    "GeneoSpmvCreate": (C.c_int, [C.POINTER(GeneoCsr), C.POINTER(C.c_void_p)]),
    "GeneoSpmvApply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "GeneoSpmvTime": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_dbl_p]),
    "GeneoSpmvDestroy": (C.c_int, [C.POINTER(C.c_void_p)]),
    "GeneoSpmvProfileStart": (C.c_int, [C.c_int, C.c_double]),
    "GeneoSpmvProfileStop": (C.c_int, [c_dbl_p, c_dbl_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "GeneoKernelProfileStart": (C.c_int, [C.c_int, C.c_double]),
    "GeneoDeviceMemInfo": (C.c_int, [c_dbl_p] * 6 + [C.c_int]),
    "GeneoDriverMain": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "GeneoKernelProfileStop": (C.c_int, []),
    "GeneoKernelProfileGet": (C.c_int, [C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "GeneoSpmmApply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "GeneoSpmmTime": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                      C.c_int, c_dbl_p]),
    "GeneoSpmmFused": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_double]),
    "GeneoSpmmDualTest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "GeneoSpmmDualResidualTest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  c_int_p, c_dbl_p, c_dbl_p]),
    "GeneoSpmvFusedSingle": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_double]),
    "GeneoSetParReduceMin": (C.c_int, [C.c_int]),
    "GeneoTestLobpcgUpdate": (C.c_int, [C.c_int, c_int_p] + [c_dbl_p] * 11),
    "GeneoBlockKernel": (C.c_int, [C.c_int, C.c_int, c_int_p, c_dbl_p, C.c_int, c_dbl_p, C.c_int, c_dbl_p,
                                   C.c_int, c_dbl_p]),
}


def bind(path):
    """dlopen `path` and attach the prototypes of include/geneo_c.h.  Raises if a symbol is missing."""
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load():
    """The product library.  Fails loudly when the HIP build is absent -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libgeneopc.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = bind(LIB_PATH)
    backend = lib.GeneoBackendName().decode()
    if backend != "hip-gfx950":
        raise RuntimeError("libgeneopc.so was not built with the HIP backend (got %r)" % backend)
    _lib = lib
    return lib
