#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per
pass): a calibration kernel with a known byte count in the SAME access width as the SpMV (8-byte lanes:
k_axpby, y = a x + b y reads 16 B and writes 8 B per element) followed by the SpMV kernels on the bench
matrix.  scripts/pmc_report.py turns the two CSVs into per-launch traffic with the calibration applied."""
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
    lib = _lib.load()
    doms = [decomp.decompose_grid_domain(n, 3, (2, 2, 2), 2, s) for s in range(8)]
    a = sp.block_diag([d.a_dir for d in doms], format="csr")
    x = DeviceVector.from_host(lib, np.random.default_rng(0).random(a.shape[0]))
    y = DeviceVector(lib, a.shape[0])
    # calibration: 40 M-element vectors (320 MB each, beyond the 256 MiB Infinity Cache)
    big = 40_000_000
    u = DeviceVector.from_host(lib, np.ones(big))
    v = DeviceVector.from_host(lib, np.ones(big))
    import ctypes as C
    hs = Spmv(a, lib)
    for kind in (1, 0):
        lib.GeneoSetSpmvKind(kind)
        h = Spmv(a, lib)
        for _ in range(10):
            # evict: stream the two big vectors through the caches between SpMV launches
            lib.GeneoTestAxpby(u.ptr, v.ptr, C.c_double(0.5), C.c_double(0.5), big)
            h.apply(x)
        h.destroy()
    lib.GeneoDeviceSync()
    print("rows", a.shape[0], "nnz", a.nnz, "algorithmic_bytes", hs.algorithmic_bytes(), "calib_elems", big)


if __name__ == "__main__":
    main()
