#!/usr/bin/env python3
"""bench.py -- GenEO-PCG setup + solve on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full pass of the hot path on the resident subdomain matrices:
    KSPSetUp  (setUpGenEOPC: level-1 set-up, LOBPCG eigensolves, Z, E = Z^T A Z, factorisation)
  + KSPSolve  (PCG, every iteration = MATIS SpMV + applyGenEOPC with coarse correction)
Workload at N = 1 = BASELINE.json configs[1] size (126^3 = 2.0 M DoF 3-D Laplacian of the reference's
tst/laplacian generator, 7-point) split into 8 overlapping subdomains on the one GPU so that the whole
two-level method (eigensolves, coarse space, RAS/ASM apply) runs; weak scaling: N GPUs hold
N * 126^3 DoF (8 subdomains per GPU), halo + all-reduce over RCCL.

`value` = SpMV GB/s (the metric's bandwidth figure): algorithmic bytes of the CSR SpMV launches issued
inside the timed steps / their HIP-event time on the launch stream, summed over ranks.  Setup and solve
seconds are reported next to it (`setup_s`, `solve_s`, `ms_per_step` = their sum).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)


def rank_grid(n_ranks):
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(n_ranks, (n_ranks, 1, 1))


def build_problem(args, rank, size):
    from geneo4petsc_amd import decomp
    n = args.n if args.n else int(round((args.n_per_gpu ** 3 * size) ** (1.0 / 3.0)))
    rg = rank_grid(size)
    parts = tuple(2 * r for r in rg)                 # 8 subdomains per GPU
    nb = parts[0] * parts[1] * parts[2]
    # subdomain (bi,bj,bk) -> rank of its 2x2x2 block
    sub_rank = np.zeros(nb, dtype=np.int64)
    for bk in range(parts[2]):
        for bj in range(parts[1]):
            for bi in range(parts[0]):
                s = bi + parts[0] * (bj + parts[1] * bk)
                sub_rank[s] = (bi // 2) + rg[0] * ((bj // 2) + rg[1] * (bk // 2))
    my = [s for s in range(nb) if sub_rank[s] == rank]
    doms = [decomp.decompose_grid_domain(n, 3, parts, args.overlap, s) for s in my]
    plan = decomp.grid_rank_plan(n, 3, parts, args.overlap, sub_rank, rank, size, doms)
    # b = A (1, 2, ..., N) (driver:820-831) on the owned rows
    npart_of = lambda gid: ((gid % n) * parts[0]) // n + parts[0] * ((((gid // n) % n) * parts[1]) // n
                                                                     + parts[1] * (((gid // (n * n)) * parts[2]) // n))
    b = np.zeros(len(plan.owned))
    for d in doms:
        rows = d.a_dir @ (d.l2g.astype(np.float64) + 1.0)
        sel = npart_of(d.l2g) == d.gid
        b[np.searchsorted(plan.owned, d.l2g[sel])] = rows[sel]
    return n, nb, doms, plan, b


def cpu_baseline(args, doms):
    """The oracle's CPU kernels / algorithm on this box's host cores (rank 0, bounded sample)."""
    import scipy.sparse as sp
    out = {"kind": "port", "unit": "GB/s"}
    # (1) CSR SpMV, C + OpenMP restatement of MatMult_SeqAIJ, on the same block-diagonal local matrix
    so = os.path.join(ROOT, "oracle", "liboracle_kernels.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "csrc")])
    lib = C.CDLL(so)
    lib.oracle_num_threads.restype = C.c_int
    a = sp.block_diag([d.a_dir for d in doms], format="csr")
    rp, col, val = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)
    x = np.random.default_rng(0).random(a.shape[0])
    y = np.zeros(a.shape[0])
    args_c = (C.c_int(a.shape[0]), rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
              val.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    lib.oracle_csr_spmv(*args_c)
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 5.0 and reps < 400:
        lib.oracle_csr_spmv(*args_c)
        reps += 1
    dt = (time.perf_counter() - t0) / max(1, reps)
    nbytes = a.nnz * 12 + (a.shape[0] + 1) * 4 + a.shape[0] * 16
    out["value"] = nbytes / dt / 1e9
    out["cores"] = int(lib.oracle_num_threads())
    out["sample"] = "CSR SpMV of the same %d-row / %d-nnz local matrix, %d repetitions (C + OpenMP)" % (
        a.shape[0], a.nnz, reps)
    # (2) the oracle's GenEO setup + PCG solve (SuperLU local solves, LAPACK/ARPACK eigensolves) on a
    #     bounded sample of the same workload: same operator and options on a smaller grid
    try:
        from geneo4petsc_amd import decomp
        from oracle import geneo_oracle as go
        ns = args.cpu_sample_n
        mesh = decomp.grid_mesh(n=ns, dim=3)
        dec = decomp.decompose(mesh, 8, None, decomp.structured_node_partition(ns, 3, (2, 2, 2)), False, args.overlap)
        am = decomp.global_matrix(mesh)
        bs = decomp.rhs_default(am)
        argv = geneo_argv(args)
        subs = [go.Subdomain(d.l2g, d.a_neu, d.mult, d.intersect) for d in dec.domains]
        t0 = time.perf_counter()
        orc = go.GenEOOracle(mesh.nbNode, subs, go.parse_options(argv)).setup(bs)
        t1 = time.perf_counter()
        res = go.solve(orc, bs, "cg", rtol=args.rtol)
        t2 = time.perf_counter()
        out["geneo_sample"] = {"grid": "%d^3 (%d DoF), 8 subdomains" % (ns, mesh.nbNode), "setup_s": t1 - t0,
                               "solve_s": t2 - t1, "iterations": res.its, "dimE": int(orc.dimE), "cores": 1}
    except Exception as e:     # the SpMV leg above is the contract; this leg is extra context
        out["geneo_sample"] = {"error": repr(e)}
    return out


def geneo_argv(args):
    return ["-geneo_lvl", args.lvl, "-geneo_tau", str(args.tau), "-geneo_cut", str(args.cut),
            "-els2_eps_tol", str(args.eps_tol), "-ksp_type", "cg", "-ksp_rtol", str(args.rtol),
            "-dls1_ksp_rtol", str(args.dls1_rtol), "-dls1_pc_type", args.dls1_pc, "-els2_pc_type", args.els2_pc] \
        + args.pc_args.split()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-per-gpu", type=int, default=126, help="grid points per side per GPU (126^3 = 2.0 M DoF)")
    ap.add_argument("--n", type=int, default=0, help="override the global grid side (e.g. 368 for the 50 M case)")
    ap.add_argument("--overlap", type=int, default=2)
    ap.add_argument("--lvl", default="SRAS,1", help="-geneo_lvl; SRAS keeps the RAS weighting and a CG-legal (symmetric) PC")
    ap.add_argument("--tau", type=float, default=0.35)
    ap.add_argument("--cut", type=int, default=20)
    ap.add_argument("--eps-tol", type=float, default=1e-3, help="reference default, geneo.cpp:658")
    ap.add_argument("--rtol", type=float, default=1e-5, help="PETSc KSP default rtol")
    ap.add_argument("--dls1-rtol", type=float, default=1e-6,
                    help="relative tolerance of the inner (local) solves; 1e-6 leaves the outer PCG untouched at rtol 1e-5 "
                         "(23 iterations and true residual 1.7814e-3 with 1e-6, 1e-8 and 1e-10 alike; 25 iterations with 1e-4)")
    ap.add_argument("--dls1-pc", default="amg", help="inner preconditioner of the local solves: amg | jacobi")
    ap.add_argument("--els2-pc", default="amg", help="LOBPCG preconditioner: amg | cheb")
    ap.add_argument("--pc-args", default="", help="further options for the PC, e.g. '-els2_amg_plain 1'")
    ap.add_argument("--cpu-sample-n", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, size))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: the GenEO hot path has no CPU fallback")
    # GENEO_BENCH_COMM=staged: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share
    # devices, gloo + host-staged halo exchange); the driver's multi-GPU runs use RCCL (backend "nccl").
    staged = os.environ.get("GENEO_BENCH_COMM") == "staged"
    if staged:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cpu" if staged else "cuda"
    if size > 1:
        import torch.distributed as dist
        if staged:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from geneo4petsc_amd import _lib
    from geneo4petsc_amd.pc import GenEOPC, DeviceVector
    lib = _lib.load()
    lib.GeneoSetStream(C.c_void_p(torch.cuda.current_stream().cuda_stream))

    t_prep = time.perf_counter()
    n, nb, doms, plan, b = build_problem(args, rank, size)
    comm = None
    if size > 1:
        from geneo4petsc_amd.comm import TorchComm, StagedComm
        comm = StagedComm(plan, lib) if staged else TorchComm(plan, torch.device("cuda", local_rank))
    prep_s = time.perf_counter() - t_prep
    argv = geneo_argv(args)
    bd = DeviceVector.from_host(lib, b)

    def make_pc():
        pc = GenEOPC(lib)
        pc.set_from_options(argv)
        pc.set_sizes(n ** 3, nb)
        if comm is not None:
            comm.attach(pc)
        for d in doms:
            pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
        return pc

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(pc):
        pc.setup(bd)
        x, its, rnorm, reason = pc.solve(bd)
        info = pc.info()
        x.free()
        return its, reason, info

    pcs = [make_pc() for _ in range(args.warmup + args.steps + 1)]
    for i in range(args.warmup):
        step(pcs[i])
        pcs[i].destroy()
    # roofline kernel = the CSR SpMV on the fine (subdomain) matrices; the small coarse-level launches of the
    # inner AMG hierarchy use the same kernel but are latency-, not bandwidth-bound: time the fine ones only
    fine_bytes = sum(d.a_dir.nnz for d in doms) * 12.0 + sum(len(d.l2g) for d in doms) * 20.0
    lib.GeneoSpmvProfileStart(4, C.c_double(0.9 * fine_bytes))
    barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(args.warmup, args.warmup + args.steps):
        last = step(pcs[i])
        if i + 1 < args.warmup + args.steps:
            pcs[i].destroy()
    barrier()
    elapsed = time.perf_counter() - t0
    ms_sum, by_sum = C.c_double(0), C.c_double(0)
    nsamp, nlaunch = C.c_longlong(0), C.c_longlong(0)
    lib.GeneoSpmvProfileStop(C.byref(ms_sum), C.byref(by_sum), C.byref(nsamp), C.byref(nlaunch))
    its, reason, info = last
    # One more, UNTIMED step with the in-situ timer off: the inner PCG chunks then replay as HIP graphs (the timer
    # needs direct launches: HIP events cannot bracket kernels inside a graph), which is how the library runs
    # outside this benchmark.  Reported as information only.
    pcs[-2].destroy()
    pcs[-1].setup(bd)
    xg, gits, _, greason = pcs[-1].solve(bd)
    ginfo = pcs[-1].info()
    # true residual || A x - b || / || b || of that solve (driver:1072-1087), owned rows, summed over the ranks
    axg = pcs[-1].matmult(xg)
    rr = axg.to_host() - b
    num, den = float(rr @ rr), float(b @ b)
    xg.free()
    axg.free()
    if dist is not None:
        tt = torch.tensor([num, den], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt)
        num, den = float(tt[0]), float(tt[1])
    true_res = (num / den) ** 0.5
    barrier()
    pc = pcs[-1]
    local = {"elapsed": elapsed, "spmv_ms": ms_sum.value, "spmv_bytes": by_sum.value, "setup": info["setupTime"],
             "solve": info["solveTime"]}
    if dist is not None:
        t = torch.tensor([elapsed, info["setupTime"], info["solveTime"]], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, setup_s, solve_s = [float(v) for v in t.tolist()]
        g = torch.tensor([by_sum.value / max(ms_sum.value, 1e-9) * 1e-6], device=red_dev, dtype=torch.float64)
        dist.all_reduce(g)                      # aggregate GB/s over ranks
        agg_gbs = float(g.item())
    else:
        setup_s, solve_s = info["setupTime"], info["solveTime"]
        agg_gbs = by_sum.value / max(ms_sum.value, 1e-9) * 1e-6
    gbs_rank = local["spmv_bytes"] / max(local["spmv_ms"], 1e-9) * 1e-6
    # HBM traffic per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, calibrated as the guide prescribes:
    # scripts/pmc_spmv.py + pmc_report.py, result committed under profiles/); only quoted when this run's
    # matrix is the profiled one
    traffic, traffic_src = None, None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_spmv_hbm_traffic_pmc.json")))
        kname = lib.GeneoSpmvKernelName().decode()
        alg = by_sum.value / max(1, nsamp.value)
        if kname in prof and abs(prof[kname]["algorithmic_bytes"] - alg) <= 0.01 * alg:
            traffic = prof[kname]["traffic_bytes_corrected"]
            traffic_src = "profiles/r01_spmv_hbm_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    except Exception:
        pass
    if rank == 0:
        out = {
            "metric": "GenEO-PCG setup+solve sec and SpMV GB/s, 3D Laplacian 50M DoF, 1/2/4/8 GPUs",
            "value": agg_gbs, "unit": "GB/s", "n_gpus": size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D 7-pt Laplacian (reference tst/laplacian generator, kappa=1, eps=1e-4), "
                                   "%d^3 = %d DoF, %d subdomains (8 per GPU), overlap %d, -geneo_lvl %s, "
                                   "-geneo_cut %d, tau %.2f, PCG rtol %.0e; N=1 is BASELINE configs[1] size (126^3)"
                                   % (n, n ** 3, nb, args.overlap, args.lvl, args.cut, args.tau, args.rtol),
                       "grid": n, "dof": n ** 3, "subdomains": nb, "overlap": args.overlap},
            "setup_s": setup_s, "solve_s": solve_s, "setup_plus_solve_s": setup_s + solve_s,
            "iterations": its, "converged": reason, "dimE": info["dimE"], "eig_iterations": info["eig_iterations"],
            "local_solve_cg_iterations": info["dls1_iterations"], "local_solves": info["dls1_solves"],
            "amg_levels": info["amg_levels"], "amg_setup_s": info["amgSetupTime"], "host_prep_s": prep_s,
            "setup_breakdown_s": {"level1_upload_and_amg": info["lvl1SetupMinvTimeLoc"],
                                  "eigensolve_lobpcg": info["lvl2SetupEigTimeLoc"],
                                  "coarse_operator_E": info["lvl2SetupETimeLoc"]},
            "untimed_step_with_hip_graphs_s": {"setup": ginfo["setupTime"], "solve": ginfo["solveTime"],
                                               "iterations": gits, "true_residual": true_res},
            "solve_breakdown_s": {"local_solves": info["lvl1ApplyMinvTimeLoc"], "coarse_Zt": info["lvl2ApplyZtTimeLoc"],
                                  "coarse_Einv": info["lvl2ApplyEinvTimeLoc"]},
            "roofline": {"bound": "hbm", "achieved": gbs_rank, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs_rank / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": lib.GeneoSpmvKernelName().decode(), "launches_timed": int(nsamp.value),
                         "launches_total": int(nlaunch.value),
                         "avg_launch_ms": ms_sum.value / max(1, nsamp.value),
                         "algorithmic_bytes_per_launch": by_sum.value / max(1, nsamp.value)},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, doms)
        print(json.dumps(out), flush=True)
    for p in pcs:
        p.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
