"""Host-side mirror of the reference's PC-shell interface (hdr/geneo.hpp, hdr/geneo_c.h) on top of
the C ABI of libgeneopc.so.  Same call order as the reference driver (src/geneo4PETSc.cpp:1328-1367):

    pc = GenEOPC()                       # PCRegister("geneo", createGenEOPC) + PCSetType
    pc.set_from_options(argv)            # PCSetFromOptions        -> setUpGenEOPCFromOptions
    pc.init(...) | pc.add_subdomain(...) # initGenEOPC / PCGenEOSetup
    pc.setup(b)                          # KSPSetUp                -> setUpGenEOPC
    x, its, rnorm, reason = pc.solve(b)  # KSPSolve (x0 from the PC, guess flagged non zero)
    y = pc.apply(x)                      # ops->apply              -> applyGenEOPC
    pc.destroy()                         # PCDestroy               -> destroyGenEOPC

All numerics run in the HIP library; numpy arrays are only staging for inputs and results.
"""
import ctypes as C

import numpy as np

from . import _lib as L

KSP_REASONS = {2: "KSP_CONVERGED_RTOL", 3: "KSP_CONVERGED_ATOL", -3: "KSP_DIVERGED_ITS", -4: "KSP_DIVERGED_DTOL",
               -8: "KSP_DIVERGED_INDEFINITE_MAT", -9: "KSP_DIVERGED_NANORINF", 0: "KSP_CONVERGED_ITERATING"}


class GenEOError(RuntimeError):
    pass


def _csr_arrays(a):
    """scipy csr_matrix or (rowptr, col, val) -> contiguous int32/int32/float64 arrays."""
    if hasattr(a, "indptr"):
        a = a.tocsr()
        a.sort_indices()
        rp, col, val = a.indptr, a.indices, a.data
    else:
        rp, col, val = a
    return (np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(col, dtype=np.int32),
            np.ascontiguousarray(val, dtype=np.float64))


def _csr_struct(arrs):
    rp, col, val = arrs
    return L.GeneoCsr(len(rp) - 1, rp.ctypes.data_as(L.c_int_p), col.ctypes.data_as(L.c_int_p),
                      val.ctypes.data_as(L.c_dbl_p))


class DeviceVector:
    """FP64 vector in HBM, allocated by the library."""

    def __init__(self, lib, n):
        self.lib, self.n = lib, int(n)
        self.ptr = lib.GeneoDeviceAlloc(max(1, self.n) * 8)
        if not self.ptr:
            raise GenEOError("device allocation failed")

    @classmethod
    def from_host(cls, lib, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        v = cls(lib, a.size)
        if a.size and lib.GeneoH2D(v.ptr, a.ctypes.data, a.size * 8):
            raise GenEOError("H2D failed")
        return v

    def to_host(self):
        out = np.empty(self.n, dtype=np.float64)
        if self.n and self.lib.GeneoD2H(out.ctypes.data, self.ptr, self.n * 8):
            raise GenEOError("D2H failed")
        return out

    def free(self):
        if self.ptr:
            self.lib.GeneoDeviceFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GenEOPC:
    def __init__(self, lib=None):
        self.lib = lib if lib is not None else L.load()
        self.h = C.c_void_p()
        if self.lib.PCCreate_GenEO(C.byref(self.h)):
            raise GenEOError("createGenEOPC failed")
        self.n_owned = None
        self._keep = []

    # -- errors / names --------------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise GenEOError(self.lib.PCGenEOGetError(self.h).decode())

    @property
    def name(self):
        return self.lib.PCGenEOGetName(self.h).decode()

    def options(self):
        """The parsed options (the public parameter fields of the reference's geneoContext)."""
        out = {}
        for kv in self.lib.PCGenEOGetOptionsString(self.h).decode().split(";"):
            k, _, v = kv.partition("=")
            if k in ("dls1_pc", "els2_pc", "ksp_type"):
                out[k] = v
            elif "." in v or "e" in v or "inf" in v:
                out[k] = float(v)
            else:
                out[k] = int(v)
        return out

    def usage(self):
        return self.lib.usageGenEO_c().decode()

    # -- options -----------------------------------------------------------------------------
    def set_from_options(self, argv):
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        self._chk(self.lib.PCSetFromOptions_GenEO(self.h, len(argv), arr))

    def set_option(self, key, value=""):
        self._chk(self.lib.PCGenEOSetOption(self.h, key.encode(), str(value).encode()))

    # -- inputs ------------------------------------------------------------------------------
    def set_sizes(self, nb_dof, nb_sub_global):
        self.N = int(nb_dof)
        self._chk(self.lib.PCGenEOSetSizes(self.h, int(nb_dof), int(nb_sub_global)))
        if self.n_owned is None:
            self.n_owned = int(nb_dof)

    def add_subdomain(self, gid, l2g, mult, a_neu, a_dir=None):
        l2g = np.ascontiguousarray(l2g, dtype=np.int32)
        mult = np.ascontiguousarray(mult, dtype=np.int32)
        an = _csr_arrays(a_neu)
        sn = _csr_struct(an)
        sd = None
        if a_dir is not None:
            ad = _csr_arrays(a_dir)
            sd = _csr_struct(ad)
        self._chk(self.lib.PCGenEOAddSubdomain(self.h, int(gid), len(l2g), l2g.ctypes.data_as(L.c_int_p),
                                               mult.ctypes.data_as(L.c_int_p), C.byref(sn),
                                               C.byref(sd) if sd is not None else None))

    def setup_from_operators(self, nbDOF, pcMap, a_local, dofMultiplicities, pcADirLoc=None, dofIntersections=None):
        """The reference's PETSc-style pair (hdr/geneo_c.h:10): KSPSetOperators(MATIS A) then
        PCGenEOSetup(pc, pcADirLoc, dofMultiplicities, dofIntersections) -- one subdomain for this rank."""
        l2g = np.ascontiguousarray(pcMap, dtype=np.int32)
        mult = np.ascontiguousarray(dofMultiplicities, dtype=np.int32)
        an = _csr_arrays(a_local)
        mat = L.GeneoMatIS(int(nbDOF), len(l2g), l2g.ctypes.data_as(L.c_int_p), _csr_struct(an))
        self.N = int(nbDOF)
        if self.n_owned is None:
            self.n_owned = int(nbDOF)
        self._chk(self.lib.PCSetOperators_GenEO(self.h, C.byref(mat)))
        sd = None
        if pcADirLoc is not None:
            ad = _csr_arrays(pcADirLoc)
            sd = _csr_struct(ad)
        inter = None
        if dofIntersections is not None:
            keep = [np.ascontiguousarray(x, dtype=np.int32) for x in dofIntersections]
            inter = (L.GeneoIS * len(keep))(*[L.GeneoIS(len(x), x.ctypes.data_as(L.c_int_p)) for x in keep])
        self._chk(self.lib.PCGenEOSetup(self.h, C.byref(sd) if sd is not None else None,
                                        L.GeneoIS(len(mult), mult.ctypes.data_as(L.c_int_p)), inter))

    def init(self, nbDOF, nbDOFLoc, pcMap, pcA, pcADirLoc, pcB, pcX0, dofIdxDomLoc, dofIdxMultLoc,
             intersectLoc=None):
        """initGenEOPC (hdr/geneo.hpp:30-35): one subdomain for this rank.  pcB: DeviceVector or None."""
        del pcX0, dofIdxDomLoc   # x0 is produced by setup; the map carries the DOF ids
        l2g = np.ascontiguousarray(pcMap, dtype=np.int32)
        mult = np.ascontiguousarray(dofIdxMultLoc, dtype=np.uint32)
        an = _csr_arrays(pcA)
        sn = _csr_struct(an)
        sd = None
        if pcADirLoc is not None:
            ad = _csr_arrays(pcADirLoc)
            sd = _csr_struct(ad)
        self.N = int(nbDOF)
        if self.n_owned is None:
            self.n_owned = int(nbDOF)
        self._chk(self.lib.initGenEOPC_c(self.h, int(nbDOF), int(nbDOFLoc), l2g.ctypes.data_as(L.c_int_p),
                                         C.byref(sn), C.byref(sd) if sd is not None else None,
                                         pcB.ptr if pcB is not None else None, None,
                                         mult.ctypes.data_as(C.POINTER(C.c_uint))))
        if intersectLoc is not None:
            self.set_intersect(getattr(self, "rank", 0), [len(x) > 0 for x in intersectLoc])

    def set_comm(self, rank, size, owned_gid, halo_gid, recv_counts, send_counts, send_idx, exchange, allreduce,
                 send_ptr, recv_ptr, red_ptr, red_capacity):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        owned_gid, halo_gid = i32(owned_gid), i32(halo_gid)
        recv_counts, send_counts, send_idx = i32(recv_counts), i32(send_counts), i32(send_idx)
        self._cb = (L.EXCHANGE_FN(exchange), L.ALLREDUCE_FN(allreduce))   # keep alive
        self._keep += [owned_gid, halo_gid, recv_counts, send_counts, send_idx]
        p = lambda a: a.ctypes.data_as(L.c_int_p)
        self._chk(self.lib.PCGenEOSetComm(self.h, rank, size, len(owned_gid), p(owned_gid), len(halo_gid),
                                          p(halo_gid), p(recv_counts), p(send_counts), p(send_idx), self._cb[0],
                                          self._cb[1], None, send_ptr, recv_ptr, red_ptr, int(red_capacity)))
        self.n_owned = len(owned_gid)
        self.rank = int(rank)

    def set_comm_width(self, width):
        """The halo buffers hold `width` vectors per exchange (PCGenEOSetCommWidth): blocked assembly of E."""
        self._chk(self.lib.PCGenEOSetCommWidth(self.h, int(width)))

    # -- PC ops ------------------------------------------------------------------------------
    def setup(self, b=None):
        """KSPSetUp -> setUpGenEOPC.  b: DeviceVector / numpy (needed only for the E-hybrid initial guess)."""
        self._b = self._as_dev(b) if b is not None else None
        if self._b is not None:
            self._chk(self.lib.PCGenEOSetRHS(self.h, self._b.ptr))
        self._chk(self.lib.PCSetUp_GenEO(self.h))

    def _as_dev(self, a):
        if isinstance(a, DeviceVector):
            return a
        return DeviceVector.from_host(self.lib, a)

    def _unary(self, fn, x):
        xd = self._as_dev(x)
        yd = DeviceVector(self.lib, xd.n)
        self._chk(fn(self.h, xd.ptr, yd.ptr))
        return yd if isinstance(x, DeviceVector) else yd.to_host()

    def apply(self, x):
        return self._unary(self.lib.PCApply_GenEO, x)

    def apply_q(self, x):
        return self._unary(self.lib.PCGenEOApplyQ, x)

    def matmult(self, x):
        return self._unary(self.lib.MatMult_GenEO, x)

    def x0(self):
        v = DeviceVector(self.lib, self.n_owned)
        self._chk(self.lib.PCGenEOGetX0(self.h, v.ptr))
        return v

    def solve(self, b, x0=None):
        """KSPSolve with the initial guess the PC produced (geneo.cpp:1601-1607) unless x0 is given."""
        bd = self._as_dev(b)
        xd = self._as_dev(x0) if x0 is not None else self.x0()
        its, reason, rnorm = C.c_int(0), C.c_int(0), C.c_double(0.0)
        self._chk(self.lib.KSPSolve_GenEO(self.h, bd.ptr, xd.ptr, C.byref(its), C.byref(rnorm), C.byref(reason)))
        x = xd if isinstance(b, DeviceVector) else xd.to_host()
        return x, its.value, rnorm.value, KSP_REASONS.get(reason.value, str(reason.value))

    def residual_history(self):
        n = self.lib.PCGenEOGetResidualHistory(self.h, None, 0)
        out = np.zeros(max(1, n))
        self.lib.PCGenEOGetResidualHistory(self.h, out.ctypes.data_as(L.c_dbl_p), n)
        return out[:n]

    # -- results -------------------------------------------------------------------------------
    def info(self):
        i = L.GeneoInfo()
        self._chk(self.lib.PCGenEOGetInfo(self.h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in L.GeneoInfo._fields_}

    def eigenvalues(self, local_sub, candidates=False):
        fn = self.lib.PCGenEOGetCandidates if candidates else self.lib.PCGenEOGetEigenvalues
        n = fn(self.h, local_sub, None, 0)
        if n < 0:
            raise GenEOError("no such subdomain")
        out = np.zeros(max(1, n))
        fn(self.h, local_sub, out.ctypes.data_as(L.c_dbl_p), n)
        return out[:n]

    def E(self):
        d = self.lib.PCGenEOGetE(self.h, None, 0)
        out = np.zeros(max(1, d * d))
        self.lib.PCGenEOGetE(self.h, out.ctypes.data_as(L.c_dbl_p), d * d)
        return out[:d * d].reshape(d, d)

    def set_intersect(self, gid, nonempty):
        """intersectLoc emptiness flags of subdomain gid (initGenEOPC, hdr/geneo.hpp:34); GenEO-2 only."""
        f = np.ascontiguousarray(nonempty, dtype=np.int32)
        self._chk(self.lib.PCGenEOSetIntersect(self.h, int(gid), len(f), f.ctypes.data_as(L.c_int_p)))

    def local_params(self):
        """(tau_loc, gamma_loc) per local subdomain (getLocalGenEOTau / Gamma, geneo.cpp:1097-1232)."""
        n = self.lib.PCGenEOGetLocalParams(self.h, None, None, 0)
        t, g = np.zeros(max(1, n)), np.zeros(max(1, n))
        self.lib.PCGenEOGetLocalParams(self.h, t.ctypes.data_as(L.c_dbl_p), g.ctypes.data_as(L.c_dbl_p), n)
        return t[:n], g[:n]

    def local_dims(self):
        n = self.lib.PCGenEOGetLocalDims(self.h, None, 0)
        out = np.zeros(max(1, n), dtype=np.int32)
        self.lib.PCGenEOGetLocalDims(self.h, out.ctypes.data_as(L.c_int_p), n)
        return out[:n]

    def destroy(self):
        if self.h:
            self.lib.PCDestroy_GenEO(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def sparse_product(a, b=None, lib=None):
    """Device sparse products of the multigrid set-up through the test hook: A @ B, or A.T when b is None.
    Returns a scipy csr_matrix, or None when a row exceeds the kernels' per-row capacity."""
    import scipy.sparse as sp
    lib = lib if lib is not None else L.load()
    aa = _csr_arrays(a)
    sa = _csr_struct(aa)
    op, ncols, sb = 1, a.shape[1], None
    if b is not None:
        ba = _csr_arrays(b)
        sb = _csr_struct(ba)
        op, ncols = 0, b.shape[1]
    nrows = a.shape[0] if b is not None else a.shape[1]
    pb = C.byref(sb) if sb is not None else None
    nnz = lib.GeneoTestSparseProduct(op, C.byref(sa), pb, ncols, None, None, None, 0)
    if nnz == -1:
        return None
    if nnz < 0:
        raise GenEOError(lib.PCGenEOGetError(None).decode())
    rp, col, val = np.zeros(nrows + 1, dtype=np.int32), np.zeros(max(1, nnz), dtype=np.int32), np.zeros(max(1, nnz))
    lib.GeneoTestSparseProduct(op, C.byref(sa), pb, ncols, rp.ctypes.data_as(L.c_int_p), col.ctypes.data_as(L.c_int_p),
                               val.ctypes.data_as(L.c_dbl_p), nnz)
    return sp.csr_matrix((val[:nnz], col[:nnz], rp), shape=(nrows, ncols if b is not None else a.shape[0]))


class Spmv:
    """Stand-alone CSR SpMV / SpMM handle (the roofline kernel of bench.py)."""

    def __init__(self, a, lib=None):
        self.lib = lib if lib is not None else L.load()
        arrs = _csr_arrays(a)
        self.n = len(arrs[0]) - 1
        self.nnz = int(arrs[0][-1])
        self.h = C.c_void_p()
        s = _csr_struct(arrs)
        if self.lib.GeneoSpmvCreate(C.byref(s), C.byref(self.h)):
            raise GenEOError(self.lib.PCGenEOGetError(None).decode())

    def apply(self, x):
        xd = x if isinstance(x, DeviceVector) else DeviceVector.from_host(self.lib, x)
        yd = DeviceVector(self.lib, self.n)
        if self.lib.GeneoSpmvApply(self.h, xd.ptr, yd.ptr):
            raise GenEOError(self.lib.PCGenEOGetError(None).decode())
        return yd if isinstance(x, DeviceVector) else yd.to_host()

    def time(self, xd, yd, reps):
        ms = C.c_double(0.0)
        if self.lib.GeneoSpmvTime(self.h, xd.ptr, yd.ptr, int(reps), C.byref(ms)):
            raise GenEOError(self.lib.PCGenEOGetError(None).decode())
        return ms.value

    def spmm(self, X, pre=None, post=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        n, m = X.shape
        xd = DeviceVector.from_host(self.lib, X.ravel())
        yd = DeviceVector(self.lib, self.n * m)
        pr = DeviceVector.from_host(self.lib, pre) if pre is not None else None
        po = DeviceVector.from_host(self.lib, post) if post is not None else None
        if self.lib.GeneoSpmmApply(self.h, xd.ptr, yd.ptr, m, pr.ptr if pr else None, po.ptr if po else None):
            raise GenEOError(self.lib.PCGenEOGetError(None).decode())
        return yd.to_host().reshape(self.n, m)

    def fused(self, epi, X=None, B=None, Z=None, dinv=None, w=0.0):
        """Multigrid epilogues fused into the SpMV / SpMM launch (GeneoSpmmFused); returns (Y, Z_out)."""
        ref = X if X is not None else B
        ref = np.asarray(ref, dtype=np.float64)
        m = 1 if ref.ndim == 1 else ref.shape[1]
        dev = lambda a: DeviceVector.from_host(self.lib, np.ascontiguousarray(a, dtype=np.float64).ravel()) if a is not None else None
        xd, bd, dd = dev(X), dev(B), dev(dinv)
        zd = dev(Z) if Z is not None else (DeviceVector(self.lib, self.n * m) if epi == 4 else None)
        yd = DeviceVector(self.lib, self.n * m)
        p = lambda v: v.ptr if v is not None else None
        if self.lib.GeneoSpmmFused(self.h, int(epi), p(xd), yd.ptr, m, p(bd), p(zd), p(dd), float(w)):
            raise GenEOError(self.lib.PCGenEOGetError(None).decode())
        shape = (self.n,) if ref.ndim == 1 else (self.n, m)
        return yd.to_host().reshape(shape), (zd.to_host().reshape(shape) if zd is not None else None)

    def fused_single(self, epi, X=None, B=None, Z=None, dinv=None, w=0.0):
        """The single-vector launches of `fused` (epi 0: A X) on the single-precision companion of the matrix."""
        dev = lambda a: DeviceVector.from_host(self.lib, np.ascontiguousarray(a, dtype=np.float64).ravel()) if a is not None else None
        xd, bd, dd = dev(X), dev(B), dev(dinv)
        zd = dev(Z) if Z is not None else (DeviceVector(self.lib, self.n) if epi == 4 else None)
        yd = DeviceVector(self.lib, self.n)
        p = lambda v: v.ptr if v is not None else None
        if self.lib.GeneoSpmvFusedSingle(self.h, int(epi), p(xd), yd.ptr, p(bd), p(zd), p(dd), float(w)):
            raise GenEOError(self.lib.PCGenEOGetError(None).decode())
        return yd.to_host(), (zd.to_host() if zd is not None else None)

    def algorithmic_bytes(self):
        """nnz*(8+4) + (n+1)*4 + n*8 (x once) + n*8 (y)."""
        return self.nnz * 12 + (self.n + 1) * 4 + self.n * 16

    def destroy(self):
        if self.h:
            self.lib.GeneoSpmvDestroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def block_kernel(kind, suboff, S, TC, lib=None, reps=0):
    """kind 0: per-subdomain Gram S^T T (kind 2: the same through the two-left-block entry) ; kind 1: per-subdomain
    S C.  Returns (result, ms_avg)."""
    lib = lib if lib is not None else L.load()
    suboff = np.ascontiguousarray(suboff, dtype=np.int32)
    nsub = len(suboff) - 1
    S = np.ascontiguousarray(S, dtype=np.float64)
    TC = np.ascontiguousarray(TC, dtype=np.float64)
    n, p = S.shape
    if kind in (0, 2):
        q = TC.shape[1]
        out = np.zeros((nsub, p, q))
    else:
        q = TC.shape[2]
        out = np.zeros((n, q))
    ms = C.c_double(0.0)
    rc = lib.GeneoBlockKernel(kind, nsub, suboff.ctypes.data_as(L.c_int_p), S.ctypes.data_as(L.c_dbl_p), p,
                              TC.ctypes.data_as(L.c_dbl_p), q, out.ctypes.data_as(L.c_dbl_p), int(reps),
                              C.byref(ms))
    if rc:
        raise GenEOError(lib.PCGenEOGetError(None).decode())
    return out, ms.value
