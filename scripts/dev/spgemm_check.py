"""Device SpGEMM on a multigrid-like chain (A 7-point, P smoothed aggregation, R = P^T, A P, R A P, next level) against
scipy: pattern and values.  python scripts/dev/spgemm_check.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from geneo4petsc_amd import _lib
from geneo4petsc_amd.pc import sparse_product
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
e = np.ones(n)
t = sp.diags([-e[:-1], 2.0001 * e, -e[:-1]], [-1, 0, 1])
i = sp.identity(n)
a = (sp.kron(sp.kron(t, i), i) + sp.kron(sp.kron(i, t), i) + sp.kron(sp.kron(i, i), t)).tocsr()
lvl = 0
while a.shape[0] > 300:
    N = a.shape[0]
    # aggregates: greedy over rows in order (structural), like amg.cpp phase 1 + leftovers joining the first aggregated neighbour
    agg = -np.ones(N, dtype=np.int64); na = 0
    ip, ix = a.indptr, a.indices
    for r in range(N):
        nb = ix[ip[r]:ip[r + 1]]
        if agg[r] < 0 and (agg[nb] < 0).all():
            agg[nb] = na; agg[r] = na; na += 1
    for r in range(N):
        if agg[r] < 0:
            nb = ix[ip[r]:ip[r + 1]]; c = agg[nb][agg[nb] >= 0]
            agg[r] = c[0] if len(c) else na
            if not len(c): na += 1
    p0 = sp.csr_matrix((np.ones(N), (np.arange(N), agg)), shape=(N, na))
    dinv = 1.0 / a.diagonal()
    P = (p0 - 0.6 * sp.diags(dinv) @ (a @ p0)).tocsr(); P.sort_indices()
    R = P.T.tocsr(); R.sort_indices()
    for name, x, y in (("A*P0", a, p0), ("A*P", a, P), ("R*(AP)", R, (a @ P).tocsr())):
        y = y.tocsr(); y.sort_indices()
        ref = (x @ y).tocsr(); ref.sort_indices()
        # scipy drops nothing structurally in csr @ csr (explicit zeros kept)
        got = sparse_product(x, y, lib)
        if got is None:
            print(lvl, name, "device product refused (row capacity)"); continue
        got.sort_indices()
        same = got.nnz == ref.nnz and np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
        err = np.abs(got.data - ref.data).max() / np.abs(ref.data).max() if same else float("nan")
        print("level %d %-7s rows %7d nnz device %9d scipy %9d pattern %s max rel err %.2e max row %d" % (lvl, name, x.shape[0], got.nnz, ref.nnz, same, err, np.diff(ref.indptr).max()))
    a = (R @ (a @ P)).tocsr(); a.sort_indices()
    lvl += 1
