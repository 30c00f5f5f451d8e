"""TEST INFRASTRUCTURE ONLY -- oracle restatement of the GenEO PC (src/geneo.cpp) and of the
PETSc Krylov loop that calls it.

The reference wires PETSc/SLEPc/MUMPS/ARPACK together; none of them is vendored, so the
mathematics is restated with the same third-party *algorithms* available here:

  MUMPS LU of A_Dir / A_Rob / E (geneo.cpp:94-124)        -> scipy SuperLU (exact sparse LU)
  MUMPS LDL^T inertia of A - tau*B (geneo.cpp:452-560)    -> eigenvalue sign count (dense eigvalsh
                                                             for small n, SuperLU diagonal signs else)
  SLEPc EPS "arpack", GHEP, shift-invert sigma=0, tol 1e-3 (geneo.cpp:626-744)
                                                          -> scipy eigsh(sigma=0, M=B) = ARPACK
                                                             dsaupd mode 3 (the same Fortran code)
  PETSc KSPCG / KSPGMRES + KSPConvergedDefault            -> restated below (ksp_cg, ksp_gmres)

Function <-> reference map
  GenEOOracle.setup                 setUpGenEOPC            geneo.cpp:1672-1843
  _robin                            createRobinMatrix       geneo.cpp:1613-1670
  _partition_of_unity               createPartitionOfUnity  geneo.cpp:965-1000
  _estimate_nev                     estimateNumberOfEigenValues / getInertia  geneo.cpp:452-560
  _eigen_local_problem              eigenLocalProblem / eigenLocalSolve       geneo.cpp:626-963
  _local_tau / _local_gamma         getLocalGenEOTau / Gamma  geneo.cpp:1097-1232
  _build_coarse_space               buildCoarseSpaceWithGenEO geneo.cpp:1234-1366
  apply_q                           applyQ                  geneo.cpp:1435-1542
  apply                             applyGenEOPC            geneo.cpp:2051-2098
  _apply_level1 / _project          applyLevel1 / projectOnFineSpace  geneo.cpp:1902-2038
  matmult                           MatMult(MATIS)          driver:755-757
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

FLT_EPSILON = float(np.finfo(np.float32).eps)
DBL_EPSILON = float(np.finfo(np.float64).eps)


# ------------------------------------------------------------------------------- options
@dataclass
class GenEOOptions:
    """Defaults = createGenEOPC, geneo.cpp:2649-2662; parsing = setUpGenEOPCFromOptions :2329-2514."""
    lvl1ASM: bool = True
    lvl1RAS: bool = False
    lvl1SRAS: bool = False
    lvl1ORAS: bool = False
    lvl2: int = 1
    hybrid: bool = False
    effHybrid: bool = False
    optim: float = 0.0
    tau: float = 0.1
    gamma: float = 10.0
    cst: bool = False
    cut: int = -1
    noSyl: bool = False
    offload: bool = False
    # inner-solver knobs the reference forwards to SLEPc through the -els2_ prefix
    eps_tol: float = 1.0e-3      # EPSSetTolerances, geneo.cpp:658
    eps_nev: int = 1             # SLEPc default nev when no inertia estimate is available
    eps_max_it: int = 0          # 0 = library default

    @property
    def name(self) -> str:
        """buildGenEOName, geneo.cpp:2245-2268."""
        n = "geneo" + str(self.lvl2)
        if self.hybrid:
            n += "E" if self.effHybrid else "H"
        l1 = ""
        if self.lvl1ASM:
            l1 = "ASM"
        if self.lvl1RAS:
            l1 = "RAS"
        if self.lvl1SRAS:
            l1 = "SRAS"
        if self.lvl1ORAS:
            l1 = "ORAS"
        if self.lvl1SRAS and self.lvl1ORAS:
            l1 = "SORAS"
        return n + l1


def parse_options(args: List[str]) -> GenEOOptions:
    """-geneo_* command line, same spellings and validation as geneo.cpp:2344-2488."""
    o = GenEOOptions()
    i = 0
    while i < len(args):
        a = args[i]
        nxt = args[i + 1] if i + 1 < len(args) else None
        if a == "-geneo_lvl":
            parts = nxt.split(",")
            if len(parts) != 2:
                raise ValueError("invalid option -geneo_lvl")
            l1, l2 = parts
            # a repeated -geneo_lvl replaces the earlier one (PETSc's options database keeps the last value)
            o.lvl1ASM, o.lvl1RAS, o.lvl1SRAS, o.lvl1ORAS = True, False, False, False
            if l1 == "ASM":
                o.lvl1ASM = True
            elif l1 == "RAS":
                o.lvl1RAS = True
            elif l1 == "SRAS":
                o.lvl1RAS = o.lvl1SRAS = True
            elif l1 == "ORAS":
                o.lvl1RAS = o.lvl1ORAS = True
            elif l1 == "SORAS":
                o.lvl1RAS = o.lvl1SRAS = o.lvl1ORAS = True
            else:
                raise ValueError("invalid option -geneo_lvl, unknown " + l1)
            table = {"0": (0, False, False), "1": (1, False, False), "H1": (1, True, False),
                     "E1": (1, True, True), "2": (2, False, False), "H2": (2, True, False),
                     "E2": (2, True, True)}
            if l2 not in table:
                raise ValueError("invalid option -geneo_lvl, unknown " + l2)
            o.lvl2, o.hybrid, o.effHybrid = table[l2]
            i += 2
        elif a == "-geneo_optim":
            o.optim = float(nxt); i += 2
        elif a == "-geneo_tau":
            o.tau = float(nxt); i += 2
        elif a == "-geneo_gamma":
            o.gamma = float(nxt); i += 2
        elif a == "-geneo_cut":
            o.cut = int(nxt); i += 2
        elif a == "-geneo_cst":
            o.cst = True; i += 1
        elif a == "-geneo_no_syl":
            o.noSyl = True; i += 1
        elif a == "-geneo_offload":
            o.offload = True; i += 1
        elif a == "-els2_eps_tol":
            o.eps_tol = float(nxt); i += 2
        elif a == "-els2_eps_nev":
            o.eps_nev = int(nxt); i += 2
        elif a == "-els2_eps_max_it":
            o.eps_max_it = int(nxt); i += 2
        else:
            i += 1
    if o.lvl2 >= 1 and o.tau <= 0.0:
        raise ValueError("GenEO preconditioner: tau must be > 0.")
    if o.lvl2 >= 1 and o.tau >= 1.0:
        raise ValueError("GenEO preconditioner: tau must be < 1.")
    if o.lvl2 >= 2 and o.gamma <= 1.0:
        raise ValueError("GenEO preconditioner: gamma must be > 1.")
    return o


# ------------------------------------------------------------------------------- subdomain
@dataclass
class Subdomain:
    """What one MPI rank hands to initGenEOPC (hdr/geneo.hpp:30-35)."""
    l2g: np.ndarray                 # dofIdxDomLoc: ascending global ids
    a_neu: sp.csr_matrix            # MATIS local matrix
    mult: np.ndarray                # dofIdxMultLoc
    intersect: Optional[List[np.ndarray]] = None   # intersectLoc[q]
    a_dir: Optional[sp.csr_matrix] = None          # optional pcADirLoc


class _PermutedLU:
    """Exact sparse LU of P A P^T for a GIVEN fill-reducing permutation (SuperLU told to keep it: NATURAL column order,
    diagonal pivots), with solve() in the original numbering.  Same factorisation as spla.splu(A) up to the elimination
    order -- i.e. up to rounding --, at a fraction of the fill when the permutation is a geometric nested dissection of a
    structured subdomain (67^3 rows: COLAMD does not fit this container's memory eight times over, this does).  Only used
    when GenEOOracle.fill_perms is set (the headline-grid goldens, tests/golden/make_headline_goldens.py)."""

    def __init__(self, m, perm):
        self.perm = np.asarray(perm)
        mp = m.tocsr()[self.perm][:, self.perm].tocsc()
        self.lu = spla.splu(mp, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))

    def solve(self, b):
        x = np.empty_like(b, dtype=np.float64)
        x[self.perm] = self.lu.solve(np.asarray(b, dtype=np.float64)[self.perm])
        return x


def nested_dissection_box(dims) -> np.ndarray:
    """Geometric nested dissection of an nx x ny x nz box in lexicographic numbering (x fastest): the two halves first,
    the separating plane (orthogonal to the longest side) last, recursively; boxes of <= 64 nodes in natural order.
    Returns perm with perm[new] = old."""
    nx, ny, nz = [int(d) for d in dims]
    idx = np.arange(nx * ny * nz).reshape(nz, ny, nx)
    out = []

    def rec(block):
        if block.size <= 64:
            out.append(block.ravel())
            return
        ax = int(np.argmax(block.shape))
        mid = block.shape[ax] // 2
        sl = [slice(None)] * 3
        sl[ax] = slice(0, mid)
        rec(block[tuple(sl)])
        sl[ax] = slice(mid + 1, None)
        rec(block[tuple(sl)])
        sl[ax] = slice(mid, mid + 1)
        out.append(block[tuple(sl)].ravel())

    rec(idx)
    return np.concatenate(out)


def nd_perm_for_grid_subdomain(l2g, n, dim=3) -> np.ndarray:
    """nested-dissection permutation of a subdomain of the n^dim grid: the dissection of its bounding box restricted to
    the nodes the subdomain holds (a block of the structured partition plus overlap layers grown by element adjacency is a
    box with stepped faces; planes of the box still separate it).  perm[new] = local index (ascending global ids)."""
    g = np.asarray(l2g, dtype=np.int64)
    c = [(g // n ** a) % n for a in range(3)] if dim == 3 else [g % n, g // n, np.zeros_like(g)]
    lo = [int(ci.min()) for ci in c]
    dims = [int(ci.max()) - l + 1 for ci, l in zip(c, lo)]
    box = (c[0] - lo[0]) + dims[0] * ((c[1] - lo[1]) + dims[1] * (c[2] - lo[2]))
    local_of = np.full(dims[0] * dims[1] * dims[2], -1, dtype=np.int64)
    local_of[box] = np.arange(len(g))
    perm = local_of[nested_dissection_box(dims)]
    perm = perm[perm >= 0]
    assert len(perm) == len(g)
    return perm


def _inertia_negative_count(m: sp.csr_matrix, perm=None) -> tuple:
    """MatGetInertia of the LDL^T factor (geneo.cpp:491): (#neg, #zero, #pos) eigenvalues."""
    n = m.shape[0]
    if perm is not None and n > 1500:
        lu = _PermutedLU(m, perm).lu
        if np.array_equal(lu.perm_r, np.arange(n)):
            d = lu.U.diagonal()
            return int(np.sum(d < 0)), int(np.sum(d == 0)), int(np.sum(d > 0))
    if n <= 1500:
        w = sla.eigvalsh(m.toarray())
        tol = 1e-13 * max(1.0, float(np.max(np.abs(w))))
        return int(np.sum(w < -tol)), int(np.sum(np.abs(w) <= tol)), int(np.sum(w > tol))
    lu = spla.splu(m.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                   options=dict(SymmetricMode=True))
    if not np.array_equal(lu.perm_r, lu.perm_c):  # pivoted: signs of U no longer give the inertia
        w = spla.eigsh(m, k=min(n - 1, 64), sigma=0.0, which="LM", return_eigenvectors=False)
        return int(np.sum(w < 0)), 0, n - int(np.sum(w < 0))
    d = lu.U.diagonal()
    return int(np.sum(d < 0)), int(np.sum(d == 0)), int(np.sum(d > 0))


_FORK_ORACLE = None


def _fork_coarse_worker(p):
    """worker of GenEOOracle._build_coarse_space (forked: the oracle is inherited, only results travel back)"""
    orc = _FORK_ORACLE
    nic0, est0 = orc.nicolaidesLoc[p], orc.estimDimELoc[p]
    vals, vecs = orc._coarse_for_subdomain(p)
    return vals, vecs, orc.nicolaidesLoc[p] - nic0, orc.candidates[p], orc.tauLoc[p], orc.gammaLoc[p], orc.estimDimELoc[p] - est0


class GenEOOracle:
    """All ranks of the reference in one process: subs[p] is what rank p would hold."""

    def __init__(self, n_global: int, subs: List[Subdomain], opts: GenEOOptions):
        self.N = n_global
        self.subs = subs
        self.o = opts
        self.P = len(subs)
        # public counters mirroring geneoContext (hdr/geneo.hpp:96-99), per rank
        self.estimDimELoc = [0] * self.P
        self.realDimELoc = [0] * self.P
        self.nicolaidesLoc = [0] * self.P
        self.tauLoc = [-1.0] * self.P
        self.gammaLoc = [-1.0] * self.P
        self.eigvals = [None] * self.P      # eigenvalues of the vectors kept in Z_i
        self.candidates = [None] * self.P   # all converged eigenvalues (before the tau filter)
        self.Z = None
        self.E = None
        self.x0 = None
        self.dense_limit = 1500
        self.arpack_seed = 20181  # start vector of the literal-mode ARPACK runs (see _eigen_solve)
        self.exact_eigs = False   # pencils above dense_limit: ARPACK at -els2_eps_tol (the reference's call) | certified exact
        self.fill_perms = None    # per subdomain: fill-reducing permutation for every sparse LU of that subdomain (_PermutedLU)

    # -- operators -------------------------------------------------------------------
    def matmult(self, x: np.ndarray) -> np.ndarray:
        """MATIS MatMult: y = sum_i R_i^T (A_Neu,i (R_i x)), summed in rank order."""
        y = np.zeros(self.N)
        for s in self.subs:
            np.add.at(y, s.l2g, s.a_neu @ x[s.l2g])
        return y

    def global_matrix(self) -> sp.csr_matrix:
        a = sp.csr_matrix((self.N, self.N))
        for s in self.subs:
            n = len(s.l2g)
            r = sp.csr_matrix((np.ones(n), (np.arange(n), s.l2g)), shape=(n, self.N))
            a = a + r.T @ s.a_neu @ r
        a = a.tocsr()
        a.sort_indices()
        return a

    # -- setup -----------------------------------------------------------------------
    def setup(self, b: Optional[np.ndarray] = None):
        o = self.o
        self.A = self.global_matrix()                                # geneo.cpp:1692
        self.a_dir, self.a_rob, self.D, self.lu1 = [], [], [], []
        for s in self.subs:
            ad = s.a_dir
            if ad is None:                                           # geneo.cpp:1697-1705
                ad = self.A[s.l2g][:, s.l2g].tocsr()
            self.a_dir.append(ad)
            self.a_rob.append(self._robin(s, ad))                   # geneo.cpp:1720
            need_pou = bool(o.lvl2) or o.lvl1RAS                     # geneo.cpp:1725
            self.D.append(1.0 / s.mult.astype(np.float64) if need_pou else None)
        for p in range(self.P):                                      # setUpLevel1, geneo.cpp:126-160
            m = self.a_rob[p] if o.lvl1ORAS else self.a_dir[p]
            self.lu1.append(self._lu(p, m))
        if o.lvl2:
            self._build_coarse_space()                               # setUpLevel2, geneo.cpp:1544
            if b is None:
                b = np.zeros(self.N)
            self.x0 = self.apply_q(b) if o.effHybrid else np.zeros(self.N)   # :1601-1607
        else:
            self.x0 = np.zeros(self.N)
        return self

    def _lu(self, p, m):
        """exact LU of a matrix of subdomain p (MUMPS in the reference, geneo.cpp:94-124; SuperLU here)"""
        if self.fill_perms is not None:
            return _PermutedLU(m, self.fill_perms[p])
        return spla.splu(m.tocsc())

    def _robin(self, s: Subdomain, a_dir):
        """createRobinMatrix, geneo.cpp:1613-1670 (dense border block incl. explicit zeros)."""
        o = self.o
        if not o.lvl1ORAS:
            return None
        if abs(o.optim) <= DBL_EPSILON:
            return a_dir.copy()
        bd = np.nonzero(s.mult > 1)[0]
        if len(bd) == 0:
            return a_dir.copy()
        blk = s.a_neu[bd][:, bd].toarray()
        n = len(s.l2g)
        rr, cc = np.meshgrid(bd, bd, indexing="ij")
        sub = sp.csr_matrix((blk.ravel(), (rr.ravel(), cc.ravel())), shape=(n, n))
        return (a_dir + o.optim * sub).tocsr()

    def _estimate_nev(self, p, a, b, param, pb) -> int:
        """estimateNumberOfEigenValues, geneo.cpp:502-560."""
        syl = (a - param * b).tocsr()
        neg, _null, pos = _inertia_negative_count(syl, None if self.fill_perms is None else self.fill_perms[p])
        est = neg if pb == "tau" else pos
        n = len(self.subs[p].l2g)
        if est > n:
            est = n
        if self.o.cut > 0 and est > self.o.cut:
            est = self.o.cut
        self.estimDimELoc[p] += est
        return est

    def _eigen_solve(self, a, b, nev, pb, p=None):
        """eigenLocalSolve, geneo.cpp:626-744 (without the tau/gamma filter)."""
        n = a.shape[0]
        tol = self.o.eps_tol
        # Dense LAPACK ground truth whenever it is affordable: single-vector Lanczos (ARPACK) can miss
        # copies of (near-)degenerate eigenvalues, which symmetric subdomains produce in clusters;
        # the checker must not inherit that solver quirk.  ARPACK is kept for larger pencils.
        dense = n <= self.dense_limit or nev >= n - 1
        if dense:
            w, v = sla.eigh(a.toarray(), b.toarray())
            if pb == "tau":      # target 0, TARGET_MAGNITUDE
                order = np.argsort(np.abs(w), kind="stable")
            else:                # LARGEST_MAGNITUDE
                order = np.argsort(-np.abs(w), kind="stable")
            order = order[:min(nev, n)]
            return w[order], v[:, order]
        perm = None if (self.fill_perms is None or p is None) else self.fill_perms[p]
        if self.exact_eigs and pb == "tau":
            return self._eigen_solve_complete(a, b, nev, perm)
        # fixed start vector: ARPACK's own random start continues one Fortran RNG stream across calls, so the same
        # pencil would give different bases depending on what ran before (SLEPc seeds its EPS start vector once)
        kw = {"v0": np.random.default_rng(self.arpack_seed).random(n) + 0.5}
        if self.o.eps_max_it > 0:
            kw["maxiter"] = self.o.eps_max_it * n
        ncv = min(n - 1, max(2 * nev + 1, 20))
        if pb == "tau":
            if perm is not None:      # shift-invert through the permuted LU of A (sigma = 0): same operator, less fill
                lu = _PermutedLU(a, perm)
                kw["OPinv"] = spla.LinearOperator(a.shape, matvec=lu.solve, dtype=np.float64)
            w, v = spla.eigsh(a.tocsc(), k=nev, M=b.tocsc(), sigma=0.0, which="LM", tol=tol, ncv=ncv, **kw)
            order = np.argsort(np.abs(w), kind="stable")
        else:
            w, v = spla.eigsh(a.tocsc(), k=nev, M=b.tocsc(), which="LM", tol=tol, ncv=ncv, **kw)
            order = np.argsort(-np.abs(w), kind="stable")
        return w[order], v[:, order]

    def _eigen_solve_complete(self, a, b, nev, perm=None):
        """The nev lowest eigenpairs of a pencil too large for LAPACK, to machine precision AND provably complete:
        ARPACK shift-invert (the solver SLEPc drives for the reference, geneo.cpp:649-663) run with tol = 0 and a
        guard of extra pairs, then Sylvester's law of inertia on A - s B (the reference's own device, geneo.cpp:452-500)
        with a shift s inside a gap above the wanted ones: the number of negative pivots must equal the number of
        computed values below s, else a copy of a multiplet was missed and the guard is widened."""
        n = a.shape[0]
        k = min(n - 2, nev + 12)
        ac, bc = a.tocsc(), b.tocsc()
        kw = {}
        if perm is not None:
            lu = _PermutedLU(a, perm)
            kw["OPinv"] = spla.LinearOperator(a.shape, matvec=lu.solve, dtype=np.float64)
        while True:
            w, v = spla.eigsh(ac, k=k, M=bc, sigma=0.0, which="LM", tol=0, ncv=min(n - 1, max(3 * k, 60)), **kw)
            order = np.argsort(np.abs(w), kind="stable")
            w, v = w[order], v[:, order]
            below = None
            for j in range(len(w) - 1, nev - 1, -1):        # a shift inside a relative gap of at least 1e-6
                if w[j] - w[j - 1] > 1e-6 * abs(w[j]):
                    shift, below = 0.5 * (w[j] + w[j - 1]), j
                    break
            if below is not None:
                neg, _null, _pos = _inertia_negative_count((a - shift * b).tocsr(), perm)
                if neg == below:
                    return w[:nev], v[:, :nev]
            if k >= n - 2:
                raise RuntimeError("oracle: could not certify the lowest %d eigenvalues" % nev)
            k = min(n - 2, k + 16)

    def _eigen_local_problem(self, p, a, b, param, pb, vals, vecs):
        """eigenLocalProblem, geneo.cpp:842-963."""
        o = self.o
        nev = o.eps_nev
        if not o.noSyl:
            est = self._estimate_nev(p, a, b, param, pb)
            if est > 0:
                nev = est                                    # :861-863
        if o.cut > 0 and nev > o.cut:                        # :871-879
            nev = o.cut
        w, v = self._eigen_solve(a, b, nev, pb, p)
        self.candidates[p] = (self.candidates[p] or []) + list(w)
        my_vals, my_vecs = [], []
        for k in range(len(w)):                              # :709-722
            if pb == "tau" and w[k] > param:
                continue
            if pb == "gamma" and w[k] < param:
                continue
            my_vals.append(float(w[k]))
            my_vecs.append(v[:, k].copy())
        if pb == "tau":                                      # Nicolaides, :897-944
            if len(my_vals) > 0 and min(my_vals) >= DBL_EPSILON:
                one = np.ones(a.shape[0])
                num = float((a @ one) @ one)
                den = float((b @ one) @ one)
                if abs(num / den) <= FLT_EPSILON:
                    my_vals.append(0.0)
                    my_vecs.append(one)
                    self.nicolaidesLoc[p] += 1
        vals.extend(my_vals)
        vecs.extend(my_vecs)

    def _local_tau(self, p) -> float:
        """getLocalGenEOTau, geneo.cpp:1097-1118."""
        if self.o.cst:
            return self.o.tau
        k = int(np.max(self.subs[p].mult))
        t = k * self.o.tau
        if t >= 1.0:
            t = 0.9
        self.tauLoc[p] = t
        return t

    def _local_gamma(self, p) -> float:
        """getLocalGenEOGamma, geneo.cpp:1120-1232 (follows the code, incl. the inverted test :1143-1145)."""
        g = self.o.gamma
        if self.o.cst:
            return g
        c = np.zeros((self.P, self.P))
        for r in range(self.P):
            inter = self.subs[r].intersect
            for q in range(self.P):
                if r == q:
                    c[r, q] = 1.0
                else:
                    has = inter is not None and len(inter[q]) > 0
                    c[r, q] = 0.0 if has else 1.0
        f = 1.0 / c.sum(axis=1)
        m = c * f[:, None] * f[None, :]
        lam = sla.eigvalsh(0.5 * (m + m.T))
        lmax = lam[np.argmax(np.abs(lam))]
        g = g / lmax
        g = g * f[p] * f[p]
        if g <= 1.0:
            g = 1.1
        self.gammaLoc[p] = g
        return g

    def _coarse_for_subdomain(self, p):
        """the eigenproblems of subdomain p (buildCoarseSpaceWithGenEO, geneo.cpp:1243-1300): (eigenvalues, vectors) kept"""
        o = self.o
        s = self.subs[p]
        dm = sp.diags(self.D[p])
        b = (dm @ self.a_dir[p] @ dm).tocsr()                # :1243-1247
        vals, vecs = [], []
        if o.lvl2 == 1:
            self._eigen_local_problem(p, s.a_neu, b, o.tau, "tau", vals, vecs)
        else:
            t = self._local_tau(p)
            self._eigen_local_problem(p, s.a_neu, self.a_rob[p], t, "tau", vals, vecs)
            g = self._local_gamma(p)
            self._eigen_local_problem(p, b, self.a_rob[p], g, "gamma", vals, vecs)
        return vals, vecs

    def _build_coarse_space(self):
        """buildCoarseSpaceWithGenEO, geneo.cpp:1234-1366."""
        o = self.o
        cut_saved = o.cut
        if o.lvl2 == 2 and o.cut >= 2:                       # :1275 (each rank halves its own copy)
            o.cut = o.cut // 2
        zcols = []      # list of (p, local vector D*v)
        # self.workers > 1: the subdomains' eigenproblems on forked worker processes (they are independent: one per MPI
        # rank in the reference) -- only used by bench.py's cpu_baseline leg, which states the core count it used
        results = None
        if getattr(self, "workers", 1) > 1 and self.P > 1:
            import multiprocessing as mp
            global _FORK_ORACLE
            _FORK_ORACLE = self
            with mp.get_context("fork").Pool(min(self.workers, self.P)) as pool:
                results = pool.map(_fork_coarse_worker, range(self.P))
            _FORK_ORACLE = None
        for p, s in enumerate(self.subs):
            d = self.D[p]
            if results is not None:
                vals, vecs, nic, cand, tl, gl, est = results[p]
                self.nicolaidesLoc[p] += nic
                self.estimDimELoc[p] += est
                self.candidates[p] = cand
                self.tauLoc[p], self.gammaLoc[p] = tl, gl
            else:
                vals, vecs = self._coarse_for_subdomain(p)
            if len(vecs) == 0:                               # :1305-1314
                vals.append(0.0)
                vecs.append(np.ones(len(s.l2g)))
                self.nicolaidesLoc[p] += 1
            self.realDimELoc[p] = len(vecs)
            self.eigvals[p] = np.array(vals)
            for v in vecs:
                zcols.append((p, d * v))                     # fillZE2L :261
        o.cut = cut_saved
        dim_e = len(zcols)
        rows, cols, data = [], [], []
        for k, (p, v) in enumerate(zcols):                   # createZE2G :355-450
            rows.append(self.subs[p].l2g)
            cols.append(np.full(len(v), k))
            data.append(v)
        self.Z = sp.csr_matrix((np.concatenate(data), (np.concatenate(rows), np.concatenate(cols))),
                               shape=(self.N, dim_e))
        self.E = (self.Z.T @ self.A @ self.Z).toarray()      # createEEig :1033
        self.luE = sla.lu_factor(self.E)
        self.dimE = dim_e

    # -- apply -----------------------------------------------------------------------
    def apply_q(self, x):
        """applyQ, geneo.cpp:1435-1542: Z E^-1 Z^T x."""
        y = self.Z.T @ x
        y = sla.lu_solve(self.luE, y)
        return self.Z @ y

    def _level1_local(self, w):
        """Scatter, [D] M^-1 [D], gather-add (geneo.cpp:1983-2025)."""
        o = self.o
        out = np.zeros(self.N)
        for p, s in enumerate(self.subs):
            wl = w[s.l2g].copy()
            if o.lvl1RAS:
                wl = wl * self.D[p]
            wl = self.lu1[p].solve(wl)
            if o.lvl1SRAS:
                wl = wl * self.D[p]
            np.add.at(out, s.l2g, wl)
        return out

    def apply(self, x):
        """applyGenEOPC, geneo.cpp:2051-2098."""
        o = self.o
        y = np.zeros(self.N)
        if o.lvl2 and not o.effHybrid:
            y = self.apply_q(x)                                   # applyLevel2
        w = x.copy()
        if o.hybrid and not o.effHybrid:
            w = w - self.matmult(y)                               # (I - P^T): x - A (Q x), :1931,:1944
        w = self._level1_local(w)
        if o.hybrid:
            w = w - self.apply_q(self.matmult(w))                 # (I - P): w - Q (A w), :1935-1944
        return y + w


# --------------------------------------------------------------------------------- Krylov
@dataclass
class KSPResult:
    x: np.ndarray
    its: int
    rnorm: float
    reason: str
    history: List[float] = field(default_factory=list)


class _Conv:
    """KSPConvergedDefault (PETSc src/ksp/ksp/interface/iterativ.c), preconditioned norm, left PC."""

    def __init__(self, rtol, atol, dtol, pc_apply, b, guess_nonzero):
        self.rtol, self.atol, self.dtol = rtol, atol, dtol
        self.pc_apply, self.b, self.guess_nonzero = pc_apply, b, guess_nonzero
        self.rnorm0 = None
        self.ttol = None

    def __call__(self, it, rnorm):
        if it == 0:
            if self.guess_nonzero:
                snorm = float(np.linalg.norm(self.pc_apply(self.b)))
                if snorm == 0.0:
                    snorm = rnorm
                self.rnorm0 = snorm
            else:
                self.rnorm0 = rnorm
            self.ttol = max(self.rtol * self.rnorm0, self.atol)
        if math.isnan(rnorm) or math.isinf(rnorm):
            return "KSP_DIVERGED_NANORINF"
        if rnorm <= self.ttol:
            return "KSP_CONVERGED_ATOL" if rnorm < self.atol else "KSP_CONVERGED_RTOL"
        if rnorm >= self.dtol * self.rnorm0:
            return "KSP_DIVERGED_DTOL"
        return ""


def ksp_cg(matmult, pc_apply, b, x0, rtol=1e-5, atol=1e-50, dtol=1e5, max_it=10000,
           guess_nonzero=True) -> KSPResult:
    """KSPSolve_CG (PETSc src/ksp/ksp/impls/cg/cg.c), KSP_NORM_PRECONDITIONED, as the driver
    calls it (KSPSolve at driver:1240 with KSPSetInitialGuessNonzero TRUE at driver:1348)."""
    x = x0.copy()
    r = b - matmult(x) if guess_nonzero else b.copy()
    z = pc_apply(r)
    dp = float(np.linalg.norm(z))
    conv = _Conv(rtol, atol, dtol, pc_apply, b, guess_nonzero)
    hist = [dp]
    reason = conv(0, dp)
    if reason:
        return KSPResult(x, 0, dp, reason, hist)
    p = None
    betaold = 0.0
    its = 0
    for i in range(max_it):
        its = i + 1
        beta = float(z @ r)
        if beta == 0.0:
            return KSPResult(x, its, dp, "KSP_CONVERGED_ATOL", hist)
        if i == 0:
            p = z.copy()
        else:
            p = z + (beta / betaold) * p
        w = matmult(p)
        dpi = float(p @ w)
        betaold = beta
        if dpi <= 0.0:
            return KSPResult(x, its, dp, "KSP_DIVERGED_INDEFINITE_MAT", hist)
        a = beta / dpi
        x = x + a * p
        r = r - a * w
        z = pc_apply(r)
        dp = float(np.linalg.norm(z))
        hist.append(dp)
        reason = conv(i + 1, dp)
        if reason:
            return KSPResult(x, its, dp, reason, hist)
    return KSPResult(x, its, dp, "KSP_DIVERGED_ITS", hist)


def ksp_gmres(matmult, pc_apply, b, x0, rtol=1e-5, atol=1e-50, dtol=1e5, max_it=10000,
              restart=30, guess_nonzero=True) -> KSPResult:
    """KSPSolve_GMRES (PETSc src/ksp/ksp/impls/gmres/gmres.c): left preconditioning,
    classical Gram-Schmidt (no refinement, the PETSc default), Givens-recurrence residual."""
    x = x0.copy()
    n = len(b)
    conv = _Conv(rtol, atol, dtol, pc_apply, b, guess_nonzero)
    its = 0
    hist = []
    first = True
    res = 0.0
    while True:
        r = pc_apply(b - matmult(x)) if (guess_nonzero or not first) else pc_apply(b)
        res = float(np.linalg.norm(r))
        if first:
            hist.append(res)
            reason = conv(0, res)
            if reason:
                return KSPResult(x, 0, res, reason, hist)
            first = False
        if res == 0.0:
            return KSPResult(x, its, res, "KSP_CONVERGED_ATOL", hist)
        m = restart
        v = np.zeros((m + 1, n))
        h = np.zeros((m + 1, m))
        cs, sn = np.zeros(m), np.zeros(m)
        g = np.zeros(m + 1)
        g[0] = res
        v[0] = r / res
        k = 0
        reason = ""
        while k < m and its < max_it:
            w = pc_apply(matmult(v[k]))
            hk = v[:k + 1] @ w                 # classical Gram-Schmidt
            w = w - v[:k + 1].T @ hk
            h[:k + 1, k] = hk
            hn = float(np.linalg.norm(w))
            h[k + 1, k] = hn
            if hn != 0.0:
                v[k + 1] = w / hn
            for j in range(k):                 # apply previous rotations
                t = cs[j] * h[j, k] + sn[j] * h[j + 1, k]
                h[j + 1, k] = -sn[j] * h[j, k] + cs[j] * h[j + 1, k]
                h[j, k] = t
            den = math.hypot(h[k, k], h[k + 1, k])
            cs[k], sn[k] = h[k, k] / den, h[k + 1, k] / den
            h[k, k] = den
            h[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            res = abs(g[k + 1])
            k += 1
            its += 1
            hist.append(res)
            reason = conv(its, res)
            if reason:
                break
        yk = sla.solve_triangular(h[:k, :k], g[:k]) if k > 0 else np.zeros(0)
        x = x + v[:k].T @ yk
        if reason:
            return KSPResult(x, its, res, reason, hist)
        if its >= max_it:
            return KSPResult(x, its, res, "KSP_DIVERGED_ITS", hist)


def solve(oracle: GenEOOracle, b, ksp_type="gmres", **kw) -> KSPResult:
    """driver solve(): KSPSetUp already done (oracle.setup), x0 from the PC (geneo.cpp:1601-1607)."""
    fn = ksp_cg if ksp_type == "cg" else ksp_gmres
    return fn(oracle.matmult, oracle.apply, b, oracle.x0, **kw)
