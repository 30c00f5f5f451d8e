#!/usr/bin/env python3
"""Micro-benchmark of the 32-column SpMM on the bench workload's fine-level matrix (HIP-event timed through the C ABI).

    python scripts/spmm_bench.py [grid n = 126] [subdomain split = 2] [m = 32]

Prints one JSON line: per (ld, kernel variant) the average launch time and the ALGORITHMIC GB/s
(nnz * 12 + n * 4 + 16 * m * n bytes per launch, SURVEY.md 8d / DESIGN.md section 3).  The kernel variant is picked with
GENEO_SPMM_WPX (workgroups per XCD group of the sliced kernel; 0 = the round-1 CSR kernel)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geneo4petsc_amd import _lib, decomp                      # noqa: E402
from geneo4petsc_amd.pc import Spmv, DeviceVector             # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
    split = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    reps = int(os.environ.get("REPS", "20"))
    lib = _lib.load()
    nb = split ** 3
    if os.environ.get("ONE"):          # ONE=1: the first subdomain of the decomposition alone
        nb = 1
    doms = [decomp.decompose_grid_domain(n, 3, (split,) * 3, 2, s) for s in range(nb)]
    a = sp.block_diag([d.a_dir for d in doms], format="csr")
    rows = a.shape[0]
    # ceiling experiments (MATRIX=diag | tri | planes): the same launch on a matrix that keeps only some of the 7 gathers
    kind = os.environ.get("MATRIX", "")
    if kind:
        offs = {"diag": [0], "tri": [-1, 0, 1], "planes": [-(n // split + 4) ** 2, 0, (n // split + 4) ** 2],
                "lines": [-(n // split + 4), -1, 0, 1, (n // split + 4)]}[kind]
        a = sp.diags([np.full(rows - abs(o), 1.0 + 0.1 * i) for i, o in enumerate(offs)], offs, format="csr")
    h = Spmv(a, lib)
    by = a.nnz * 12 + rows * (4 + 16 * m)
    out = {"n": n, "subdomains": nb, "rows": rows, "nnz": int(a.nnz), "m": m, "algorithmic_bytes": by,
           "wpx": os.environ.get("GENEO_SPMM_WPX", "default"), "matrix": os.environ.get("MATRIX", "A_Dir")}
    rng = np.random.default_rng(1)
    for ld in (m, 3 * m):
        Xh = rng.random((rows, ld))
        X = DeviceVector.from_host(lib, Xh.ravel())
        Y = DeviceVector(lib, rows * ld)
        ms = C.c_double(0)
        best = 1e30
        for _ in range(3):
            rc = lib.GeneoSpmmTime(h.h, X.ptr, ld, Y.ptr, ld, m, None, None, reps, C.byref(ms))
            assert rc == 0
            best = min(best, ms.value)
        if os.environ.get("CHECK"):
            y = Y.to_host().reshape(rows, ld)[:, :m]
            ref = a @ Xh[:, :m]
            out["err_ld%d" % ld] = float(np.abs(y - ref).max() / np.abs(ref).max())
        out["ld%d" % ld] = {"ms": best, "GBs": by / best * 1e-6}
        X.free()
        Y.free()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
