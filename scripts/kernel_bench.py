#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels on the bench workload's matrices (HIP-event timed)."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geneo4petsc_amd import _lib, decomp                      # noqa: E402
from geneo4petsc_amd.pc import Spmv, DeviceVector, block_kernel   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
    lib = _lib.load()
    doms = [decomp.decompose_grid_domain(n, 3, (2, 2, 2), 2, s) for s in range(8)]
    a = sp.block_diag([d.a_dir for d in doms], format="csr")
    out = {"n": n, "rows": a.shape[0], "nnz": int(a.nnz)}
    xh = np.random.default_rng(0).random(a.shape[0])
    yref = a @ xh
    x = DeviceVector.from_host(lib, xh)
    y = DeviceVector(lib, a.shape[0])
    import ctypes as C
    big = 40_000_000
    u = DeviceVector.from_host(lib, np.ones(big))
    v = DeviceVector.from_host(lib, np.ones(big))
    for kind, name in ((0, "lds"), (21, "sell_u4"), (31, "sell_u4_nt")):
        lib.GeneoSetSpmvKind(kind)
        h = Spmv(a, lib)
        err = float(np.abs(h.apply(xh) - yref).max() / np.abs(yref).max())
        ms = min(h.time(x, y, 200) for _ in range(3))
        # cold: evict caches with a 640 MB stream between launches, time each SpMV with HIP events
        lib.GeneoSpmvProfileStart(1, C.c_double(0.0))
        for _ in range(20):
            lib.GeneoTestAxpby(u.ptr, v.ptr, C.c_double(0.5), C.c_double(0.5), big)
            h.apply(x)
        msum, bsum, ns_, nl_ = C.c_double(0), C.c_double(0), C.c_longlong(0), C.c_longlong(0)
        lib.GeneoSpmvProfileStop(C.byref(msum), C.byref(bsum), C.byref(ns_), C.byref(nl_))
        out["spmv_" + name] = {"ms_warm": ms, "GBs_warm": h.algorithmic_bytes() / ms * 1e-6,
                               "ms_cold": msum.value / max(1, ns_.value), "err": err,
                               "GBs_cold": bsum.value / max(msum.value, 1e-9) * 1e-6}
        h.destroy()
    lib.GeneoSetSpmvKind(1)
    # SpMM m = 32
    h = Spmv(a, lib)
    m = 32
    X = DeviceVector.from_host(lib, np.random.default_rng(1).random(a.shape[0] * m))
    Y = DeviceVector(lib, a.shape[0] * m)
    lib.GeneoSpmmApply(h.h, X.ptr, Y.ptr, m, None, None)
    lib.GeneoDeviceSync()
    t0 = time.perf_counter()
    for _ in range(20):
        lib.GeneoSpmmApply(h.h, X.ptr, Y.ptr, m, None, None)
    lib.GeneoDeviceSync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    by = a.nnz * 12 + a.shape[0] * (4 + 16 * m)
    out["spmm32"] = {"ms": ms, "GBs": by / ms * 1e-6}
    # Gram / block update at the LOBPCG shapes
    nrow = a.shape[0]
    suboff = np.concatenate([[0], np.cumsum([len(d.l2g) for d in doms])]).astype(np.int32)
    rng = np.random.default_rng(2)
    S = rng.random((nrow, 96))
    _, ms = block_kernel(0, suboff, S, S, lib, reps=10)
    out["gram96"] = {"ms": ms, "TFs": 2.0 * nrow * 96 * 96 / ms * 1e-9, "GBs": 2.0 * nrow * 96 * 8 / ms * 1e-6}
    Cm = rng.random((8, 96, 64))
    _, ms = block_kernel(1, suboff, S, Cm, lib, reps=10)
    out["blockmul96x64"] = {"ms": ms, "TFs": 2.0 * nrow * 96 * 64 / ms * 1e-9,
                            "GBs": nrow * (96 + 64) * 8 / ms * 1e-6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
