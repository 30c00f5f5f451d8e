#!/usr/bin/env python3
"""HBM-traffic measurement of the hot kernels with rocprofv3 PMC counters, as the guide prescribes
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE in SEPARATE passes
(`--pmc` with `--kernel-trace` only), calibrated on a stream of known size in the same run -- on gfx950 FETCH_SIZE
reports half the bytes of a wide streaming read, WRITE_SIZE is exact.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d A -- python3 scripts/pmc.py work [n] [split]
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d B -- python3 scripts/pmc.py work [n] [split]
    python3 scripts/pmc.py report A B work.json > profiles/r03_hbm_traffic_pmc.json        (scripts/gpu.sh pmc2 does all)

`work` runs, on the bench workload's fine-level matrix (block-diagonal A_Dir of the 8 subdomains): the calibration
kernel k_axpby on two 40 M-element vectors (reads 640 MB, writes 320 MB: also evicts the 256 MiB Infinity Cache between
the measured launches), the CSR SpMV, the 32-column SpMM on a contiguous block (ld 32) and inside a 96-column LOBPCG
basis (ld 96), the 96 x 96 MFMA Gram, the 96 -> 64 MFMA block update, the fused three-operand LOBPCG update and its basis-only form; it writes the algorithmic bytes of each to
work.json (argv[-1] when it ends in .json)."""
import csv
import ctypes as C
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = {"k_spmv_sell<": "spmv", "k_spmv_sell_lp<": "spmv_lp_pre", "k_spmv_sell_p8": "spmv", "k_spmm_sell": "spmm32", "k_spmm<": "spmm32_csr", "k_gram_mfma": "gram96", "k_gram_flat": "gram96",
           "k_blockmul_mfma": "blockmul96x64", "k_lobpcg_update32<1>": "lobpcg_update32_basis", "k_lobpcg_update32<3>": "lobpcg_update32"}
CALIB = 40_000_000


def work(argv):
    import scipy.sparse as sp
    from geneo4petsc_amd import _lib, decomp
    from geneo4petsc_amd.pc import Spmv, DeviceVector, block_kernel
    out_json = argv[-1] if argv and argv[-1].endswith(".json") else None
    nums = [int(a) for a in argv if a.isdigit()]
    n = nums[0] if nums else 126
    split = nums[1] if len(nums) > 1 else 2
    lib = _lib.load()
    # a third number: ONLY that subdomain of the decomposition (work 368 2 0: one 6.5 M-row subdomain of the metric's
    # configuration -- the traffic / algorithmic ratios of the kernel classes at the bench line's per-subdomain size)
    only = nums[2] if len(nums) > 2 else None
    doms = [decomp.decompose_grid_domain(n, 3, (split,) * 3, 2, s, native=True) for s in (range(split ** 3) if only is None else [only])]
    a = sp.block_diag([d.a_dir for d in doms], format="csr") if len(doms) > 1 else doms[0].a_dir.tocsr()
    rows, m = a.shape[0], 32
    h = Spmv(a, lib)
    u, v = DeviceVector.from_host(lib, np.ones(CALIB)), DeviceVector.from_host(lib, np.ones(CALIB))
    evict = lambda: lib.GeneoTestAxpby(u.ptr, v.ptr, C.c_double(0.5), C.c_double(0.5), CALIB)
    x = DeviceVector.from_host(lib, np.random.default_rng(0).random(rows))
    y = DeviceVector(lib, rows)
    X = DeviceVector.from_host(lib, np.random.default_rng(1).random(rows * 96))
    Y = DeviceVector(lib, rows * 96)
    # the solver's fine matrix carries 16-bit column offsets (its single-precision companion): build them first so that
    # the SpMV measured here is the variant the local solves launch (10 B per entry)
    if os.environ.get("PMC_SPMV_32BIT_COLUMNS", "0") != "1":
        lib.GeneoSpmvFusedSingle(h.h, 0, x.ptr, y.ptr, None, None, None, C.c_double(0.0))
    for _ in range(6):
        evict()
        lib.GeneoSpmvApply(h.h, x.ptr, y.ptr)
    # the zero-guess sweep + residual of the V-cycle on the single-precision companion (EPI_PRE: b in, r out): the pass the
    # local solves launch on the fine matrix, k_spmv_sell_lp<4, 1, unsigned short, ..>
    dinv = DeviceVector.from_host(lib, np.random.default_rng(7).random(rows) + 0.5)
    for _ in range(6):
        evict()
        lib.GeneoSpmvFusedSingle(h.h, 4, None, y.ptr, x.ptr, None, dinv.ptr, C.c_double(0.7))
    for ld in (32, 96):
        for _ in range(6):
            evict()
            lib.GeneoSpmmTime(h.h, X.ptr, ld, Y.ptr, ld, m, None, None, 0, None)
    lib.GeneoDeviceSync()
    alg = {"calib_elems": CALIB, "rows": rows, "nnz": int(a.nnz),
           "spmv": a.nnz * 12 + (rows + 1) * 4 + rows * 16,
           "spmv_lp_pre": a.nnz * 6 + (rows // 64 + 1) * 4 + rows * 16,
           "spmm32": a.nnz * 12 + rows * 4 + 16 * m * rows,
           "gram96": 8 * rows * (96 + 96), "blockmul96x64": 8 * rows * (96 + 64),
           "lobpcg_update32": 8 * rows * (3 * 96 + 3 * 64 + 32), "lobpcg_update32_basis": 8 * rows * (96 + 64)}
    alg["spmm32_csr"] = alg["spmm32"]
    suboff = np.concatenate([[0], np.cumsum([len(d.l2g) for d in doms])]).astype(np.int32)
    if os.environ.get("PMC_BLOCK_KERNELS", "1") == "1":
        S = np.random.default_rng(2).random((rows, 96))
        block_kernel(0, suboff, S, S, lib, reps=3)
        block_kernel(1, suboff, S, np.random.default_rng(3).random((len(doms), 96, 64)), lib, reps=3)
        # the W rows of both Gram matrices in one pass (LOBPCG's reduced iteration: k_gram_flat<2, 3, true>) and the
        # 32 x 32 block product of the coarse-operator assembly (k_blockmul_mfma<8>)
        block_kernel(2, suboff, np.ascontiguousarray(S[:, :64]), S, lib, reps=3)
        block_kernel(1, suboff, np.ascontiguousarray(S[:, :32]), np.random.default_rng(5).random((len(doms), 32, 32)), lib, reps=3)
        # the fused LOBPCG update [X P] <- S C for S, A S, B S with the residual R = A X - B X diag(lam) (one launch)
        nd = len(doms)
        Cm = np.random.default_rng(4).random((nd, 96, 64))
        vec = np.ones((nd, 32))
        T, R = np.zeros((rows, 96)), np.zeros((rows, 32))
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        evict()
        rc = lib.GeneoTestLobpcgUpdate(nd, suboff.ctypes.data_as(C.POINTER(C.c_int)), dp(S), dp(S), dp(S), dp(Cm), dp(vec),
                                       dp(vec), dp(vec), dp(T), dp(T), dp(T), dp(R))
        assert rc == 0
        # the basis-only form of the lean iteration (AS = NULL): T = [X' P'] from S
        null = C.POINTER(C.c_double)()
        evict()
        rc = lib.GeneoTestLobpcgUpdate(nd, suboff.ctypes.data_as(C.POINTER(C.c_int)), dp(S), null, null, dp(Cm), dp(vec),
                                       null, null, dp(T), null, null, null)
        assert rc == 0
    if out_json:
        json.dump(alg, open(out_json, "w"))
    print(json.dumps(alg))


def load(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) + glob.glob(d + "*.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def report(fd, wd, alg_json):
    alg = json.load(open(alg_json))
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    mean = lambda v: sum(v) / len(v)
    import hashlib
    rep = {"kernel_source_sha16": hashlib.sha256(open(os.path.join(ROOT, "geneo4petsc_amd", "csrc", "backend_hip.hip"), "rb").read()).hexdigest()[:16],
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; counters are "
                     "in units of 1024 B; read bytes calibrated on k_axpby (16 B read + 8 B written per element, "
                     "%d elements) in the same pass" % alg["calib_elems"]}
    kax = [k for k in fetch if "k_axpby" in k]
    corr_r = corr_w = 1.0
    if kax:
        fr = mean(fetch[kax[0]]) * 1024.0
        corr_r = 16.0 * alg["calib_elems"] / fr
        rep["calibration"] = {"kernel": "k_axpby", "expected_read_bytes": 16.0 * alg["calib_elems"], "FETCH_SIZE_bytes": fr,
                              "read_correction": corr_r}
        kw = [k for k in write if "k_axpby" in k]
        if kw:
            wr = mean(write[kw[0]]) * 1024.0
            corr_w = 8.0 * alg["calib_elems"] / wr
            rep["calibration"].update({"expected_write_bytes": 8.0 * alg["calib_elems"], "WRITE_SIZE_bytes": wr,
                                       "write_correction": corr_w})
    for pat, key in KERNELS.items():
        for kname in sorted(k for k in fetch if pat in k):
            # SpMM: the first 6 launches ran on ld 32, the next 6 on ld 96
            f_all = fetch[kname]
            w_all = write.get(kname, [0.0] * len(f_all))
            groups = [("", f_all, w_all)]
            if key.startswith("spmm32") and len(f_all) == 12:
                groups = [("_ld32", f_all[:6], w_all[:6]), ("_ld96", f_all[6:], w_all[6:])]
            for suffix, fv, wv in groups:
                fr, wr = mean(fv) * 1024.0, mean(wv) * 1024.0
                short = kname.split("(")[0].replace("bk::", "").replace("void ", "").strip()
                rep[short.split("<")[0] + suffix if suffix else short] = {
                    "kernel": short, "launches": len(fv), "FETCH_SIZE_bytes_raw": fr, "WRITE_SIZE_bytes_raw": wr,
                    "read_bytes_corrected": fr * corr_r, "write_bytes_corrected": wr * corr_w,
                    "traffic_bytes_corrected": fr * corr_r + wr * corr_w, "algorithmic_bytes": alg[key], "rows": alg["rows"],
                    "traffic_over_algorithmic": (fr * corr_r + wr * corr_w) / alg[key]}
    print(json.dumps(rep, indent=1))


def mfma_report(dirs):
    """per MFMA kernel: the mean of every counter the given rocprofv3 --pmc passes collected (MfmaUtil = the derived
    metric, 100 x sum SQ_VALU_MFMA_BUSY_CYCLES / (max GRBM_GUI_ACTIVE x SIMDs), gfx94x formula as the guide notes;
    SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 = FP64 matrix flops)"""
    import hashlib
    rep = {"kernel_source_sha16": hashlib.sha256(open(os.path.join(ROOT, "geneo4petsc_amd", "csrc", "backend_hip.hip"), "rb").read()).hexdigest()[:16],
           "method": "one rocprofv3 --pmc pass per counter set with --kernel-trace only, on scripts/pmc.py work (bench matrix, LOBPCG shapes)"}
    acc = {}
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if not any(t in k for t in ("k_gram_flat", "k_gram_mfma", "k_blockmul_mfma", "k_lobpcg_update32")):
                    continue
                short = k.split("(")[0].replace("bk::", "").replace("void ", "").strip()
                acc.setdefault(short, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        row = {"launches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            row[c] = sum(v) / len(v)
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in row:
            row["fp64_mfma_flops_per_launch"] = row["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
        rep[k] = row
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "work":
        work(sys.argv[2:])
    elif sys.argv[1] == "mfma_report":
        mfma_report(sys.argv[2:])
    else:
        report(sys.argv[2], sys.argv[3], sys.argv[4])
