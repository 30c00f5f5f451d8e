#!/usr/bin/env python3
"""bench.py -- GenEO-PCG setup + solve on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

N > 1 launched plainly (no RANK in the environment): this process touches no GPU; it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` as a child
process and relays rank 0's JSON line and the exit code.  Under torchrun (RANK / WORLD_SIZE set) it is one rank per GPU.

One "step" = one full pass of the hot path on the resident subdomain matrices:
    KSPSetUp  (setUpGenEOPC: level-1 set-up, LOBPCG eigensolves, Z, E = Z^T A Z, factorisation)
  + KSPSolve  (PCG, every iteration = MATIS SpMV + applyGenEOPC with coarse correction)

Workload (3-D 7-point Laplacian of the reference's tst/laplacian generator, kappa = 1, eps = 1e-4, overlap 2,
-geneo_lvl SRAS,1 -geneo_cut 20, tau 0.35, PCG rtol 1e-5):
  --scaling strong (default): THE METRIC'S CONFIGURATION AT EVERY N (BASELINE configs[2], SURVEY 8d "config 3"): the
          368^3 = 49.8 M DoF grid in 8 subdomains (2 x 2 x 2 blocks), 8 / 4 / 2 / 1 subdomains per GPU at N = 1 / 2 / 4 / 8
          -- the reference's layout (one subdomain per MPI rank, global problem fixed, src/geneo4PETSc.cpp:604) folded
          onto fewer GPUs, so that N = 1, 2, 4, 8 is ONE curve.  On one GPU the eight 6.5 M-row eigensolves run group by
          group under a device-memory budget (-geneo_eig_mem_gb: the matrices and hierarchies of all eight stay resident,
          the LOBPCG blocks of one group at a time; results identical to the all-at-once path).  N > 1: halo exchange and
          all-reduces over RCCL (C++ transport inside libgeneopc, csrc/comm_rccl.cpp).
  --scaling weak: the round-1..3 lines -- N = 1: BASELINE configs[1] size, 126^3 = 2.0 M DoF in 8 subdomains on the one
          GPU (--n 126); N > 1: 184^3 DoF and ONE subdomain per GPU.

`value` = CSR SpMV GB/s (the metric's bandwidth figure): algorithmic bytes of the fine-level SpMV launches issued
inside the timed steps / their HIP-event time on the launch stream, summed over ranks.  Setup and solve seconds are
reported next to it (`setup_s`, `solve_s`; `ms_per_step` = wall clock of the K steps / K).  `roofline` describes the
kernel that owns the largest share of the step; `roofline.kernels` lists every hot kernel class with its share.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

T_PROCESS0 = time.perf_counter()     # bench_wall_s in the JSON line: this process from here to the print (imports included)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)
FP64_MFMA_PEAK_TFS = 78.6  # AMD MI355X datasheet FP64 matrix (the guide's table has no FP64-MFMA row); DESIGN.md section 3
KERNEL_CLASSES = [(0, "k_spmv_sell (fine-level CSR SpMV)", "hbm"), (1, "k_spmm_sell (fine-level SpMM, 32 columns)", "hbm"),
                  (2, "k_gram_flat (Rayleigh-Ritz Gram, FP64 MFMA; k_gram_mfma for other shapes)", "mfma"),
                  (3, "k_lobpcg_update32 (LOBPCG block update [X' P'] = S C, FP64 MFMA; lean iteration: S alone, the residual comes from the two-operator SpMM)", "mfma"),
                  (4, "k_spmv_sell_lp (V-cycle of the local solves: fine-level passes over the float / 16-bit-column companions)", "hbm")]


def rank_grid(n_ranks):
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(n_ranks, (n_ranks, 1, 1))


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="strong (default): the metric's 368^3 grid in 8 subdomains at every N, 8 / N subdomains per GPU; "
                         "weak: 126^3 in 8 subdomains at N = 1, 184^3 and one subdomain per GPU at N > 1 (rounds 1-3)")
    ap.add_argument("--n-per-gpu", type=int, default=0,
                    help="--scaling weak: grid points per side per GPU; 0 = 126 at N = 1 (2.0 M DoF), 184 at N > 1")
    ap.add_argument("--subdomains-per-gpu", type=int, default=0, choices=(0, 1, 2, 4, 8),
                    help="--scaling weak: 0 = 8 at N = 1, 1 at N > 1.  --scaling strong: always 8 / N")
    ap.add_argument("--n", "--grid", dest="n", type=int, default=0, help="global grid side (--scaling strong: default 368)")
    ap.add_argument("--overlap", type=int, default=2)
    ap.add_argument("--lvl", default="SRAS,1", help="-geneo_lvl; SRAS keeps the RAS weighting and a CG-legal (symmetric) PC")
    ap.add_argument("--tau", type=float, default=0.35)
    ap.add_argument("--cut", type=int, default=20)
    ap.add_argument("--eps-tol", type=float, default=1e-3, help="reference default, geneo.cpp:658")
    ap.add_argument("--rtol", type=float, default=1e-5, help="PETSc KSP default rtol")
    ap.add_argument("--dls1-rtol", type=float, default=1e-7,
                    help="relative tolerance of the inner (local) solves: 1e-7 (and 3e-7, 1e-8, 1e-10) leaves the outer PCG at "
                         "rtol 1e-5 where exact local solves put it (20^3..32^3: 22 / 25 / 24 / 26 iterations = the oracle's); "
                         "1e-6 costs 1-2 more at 24^3 and 28^3")
    ap.add_argument("--dls1-pc", default="amg", help="inner preconditioner of the local solves: amg | jacobi")
    ap.add_argument("--els2-pc", default="amg", help="LOBPCG preconditioner: amg | cheb")
    ap.add_argument("--pc-args", default="", help="further options for the PC")
    ap.add_argument("--cpu-sample-n", type=int, default=32,
                    help="grid side of the CPU-baseline / parity sample.  The oracle's reference-literal set-up (SuperLU + ARPACK "
                         "shift-invert at 1e-3, one worker process per subdomain) takes 3 s at 32^3, 57 s at 40^3, 106 s at 48^3 and "
                         "~550 s at 64^3 on 8 cores: 32^3 keeps the default run inside the 10-30 s CPU budget of the contract")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="laplacian", choices=("laplacian", "heat", "graph"),
                    help="laplacian: the metric's operator (default).  heat: BASELINE configs[3], tst/heat generator (lambda 1, "
                         "dt 0.1, --kappa K minmax: kappa = K on the middle third of every axis, K = 100 => contrast K^3 = 1e6 in "
                         "3-D), 8 subdomains; the step is timed with -geneo_lvl ASM,1 and the PCG count of plain ASM,0 is printed "
                         "beside it.  graph: BASELINE configs[4], tst/graph generator --level 2 --noGround (--graph-size 1111111 "
                         "=> 9 x 1054^2 = 10.0 M nodes), nodal partition into 8 by the library's C++ k-way partitioner (an "
                         "irregular CSR: the LDS-tiled SpMV / SpMM stress).  Both N = 1 only.")
    ap.add_argument("--kappa", type=float, default=100.0, help="heat: K of --kappa K minmax")
    ap.add_argument("--graph-size", type=int, default=1111111, help="graph: --size of tst/graph (nodes = 9 floor(sqrt(size))^2 at level 2)")
    ap.add_argument("--one-rank-of", type=int, default=0, choices=(0, 8),
                    help="8: ONE rank's share of the metric's configuration on this one GPU, uncontended -- subdomain "
                         "--rank-index of the 2x2x2 decomposition of (2 n-per-gpu)^3 (default 368^3: 186^3 local rows, overlap 2, "
                         "20 vectors) as a stand-alone problem: every communication-free phase of the set-up (SURVEY 8e: "
                         "matrices, both hierarchies, LOBPCG, Z, the local block of E) and --apply-count applications of the "
                         "preconditioner + local SpMV (what one PCG iteration costs this rank between two halo exchanges)")
    ap.add_argument("--rank-index", type=int, default=0)
    ap.add_argument("--apply-count", type=int, default=25)
    ap.add_argument("--no-anchor", action="store_true",
                    help="N > 1: skip the one-GPU anchor (184^3 in 8 subdomains on rank 0's GPU after the timed region)")
    ap.add_argument("--comm", default=os.environ.get("GENEO_BENCH_COMM", "rccl"), choices=("rccl", "torch", "staged"),
                    help="N > 1 transport: rccl = C++ ncclSend/Recv + ncclAllReduce inside libgeneopc (default); "
                         "torch = torch.distributed callbacks; staged = host-staged gloo (ranks may share a GPU)")
    return ap


def geneo_argv(args):
    return ["-geneo_lvl", args.lvl, "-geneo_tau", str(args.tau), "-geneo_cut", str(args.cut),
            "-els2_eps_tol", str(args.eps_tol), "-ksp_type", "cg", "-ksp_rtol", str(args.rtol),
            "-dls1_ksp_rtol", str(args.dls1_rtol), "-dls1_pc_type", args.dls1_pc, "-els2_pc_type", args.els2_pc] \
        + args.pc_args.split()


def workload(args, size):
    """(global grid side n, subdomain grid `parts`, subdomain -> rank map, subdomains per GPU)"""
    if args.scaling == "strong":
        if size not in (1, 2, 4, 8):
            raise SystemExit("--scaling strong: 8 subdomains on 1, 2, 4 or 8 GPUs")
        n = args.n or 368
        parts, rg = (2, 2, 2), rank_grid(size)
        sub_rank = np.zeros(8, dtype=np.int64)
        for bk in range(2):
            for bj in range(2):
                for bi in range(2):          # block (bi, bj, bk) lives on the rank whose box of the rank grid holds it
                    sub_rank[bi + 2 * (bj + 2 * bk)] = (bi * rg[0]) // 2 + rg[0] * ((bj * rg[1]) // 2 + rg[1] * ((bk * rg[2]) // 2))
        return n, parts, sub_rank, 8 // size
    spg = args.subdomains_per_gpu or (8 if size == 1 else 1)
    npg = args.n_per_gpu or (126 if size == 1 else 184)
    n = args.n if args.n else int(round((npg ** 3 * size) ** (1.0 / 3.0)))
    rg = rank_grid(size)
    f = 2 if spg == 8 else 1
    parts = tuple(f * r for r in rg)
    nb = parts[0] * parts[1] * parts[2]
    sub_rank = np.zeros(nb, dtype=np.int64)
    for bk in range(parts[2]):
        for bj in range(parts[1]):
            for bi in range(parts[0]):
                s = bi + parts[0] * (bj + parts[1] * bk)
                sub_rank[s] = (bi // f) + rg[0] * ((bj // f) + rg[1] * (bk // f))
    return n, parts, sub_rank, spg


def build_problem(args, rank, size):
    from geneo4petsc_amd import decomp
    n, parts, sub_rank, spg = workload(args, size)
    nb = len(sub_rank)
    my = [s for s in range(nb) if sub_rank[s] == rank]
    # the rank's subdomains side by side (the generator and the decomposition are C++ calls and numpy passes that release
    # the interpreter lock: 368^3 on one rank, eight 6.5 M-row subdomains: 33 s one after the other)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(my)))) as ex:
        doms = list(ex.map(lambda s: decomp.decompose_grid_domain(n, 3, parts, args.overlap, s, native=True), my))
    plan = decomp.grid_rank_plan(n, 3, parts, args.overlap, sub_rank, rank, size, doms)
    # b = A (1, 2, ..., N) (driver:820-831) on the owned rows
    npart_of = lambda gid: ((gid % n) * parts[0]) // n + parts[0] * ((((gid // n) % n) * parts[1]) // n
                                                                     + parts[1] * (((gid // (n * n)) * parts[2]) // n))
    b = np.zeros(len(plan.owned))
    for d in doms:
        rows = d.a_dir @ (d.l2g.astype(np.float64) + 1.0)
        sel = npart_of(d.l2g) == d.gid
        b[np.searchsorted(plan.owned, d.l2g[sel])] = rows[sel]
    return n, nb, spg, doms, plan, b


class _WholePlan:
    """the one-rank "plan" of an unstructured workload: this rank owns every DOF"""
    def __init__(self, n):
        self.owned = np.arange(n, dtype=np.int64)


def build_problem_unstructured(args):
    """heat / graph workloads (N = 1): (description, nDOF, nb, domains, plan, b, extra facts for the JSON line)"""
    from geneo4petsc_amd import decomp
    facts = {}
    if args.workload == "heat":
        n = args.n if args.n else (args.n_per_gpu or 126)
        gen = dict(heat=True, lbd=1.0, dt=0.1, kappa_max=args.kappa, interp="minmax")
        parts = (2, 2, 2)
        doms = [decomp.decompose_grid_domain(n, 3, parts, args.overlap, s, native=True, **gen) for s in range(8)]
        ndof = n ** 3
        # b = A (1..N) from the Dirichlet rows of the domain that owns each node (driver:820-831)
        npart = decomp.structured_node_partition(n, 3, parts)
        b = np.zeros(ndof)
        for d in doms:
            rows = d.a_dir @ (d.l2g.astype(np.float64) + 1.0)
            sel = npart[d.l2g] == d.gid
            b[d.l2g[sel]] = rows[sel]
        desc = ("tst/heat generator (reference heat.cpp: lambda 1, dt 0.1, --kappa %g minmax => contrast %.0e), %d^3 = %d DoF, "
                "8 subdomains (2x2x2), overlap %d" % (args.kappa, args.kappa ** 3, n, ndof, args.overlap))
        return desc, ndof, 8, doms, _WholePlan(ndof), b, facts
    t0 = time.perf_counter()
    mesh = decomp.graph_mesh(size=args.graph_size, level=2, no_ground=True)
    t1 = time.perf_counter()
    ep, npart, cut = decomp.partition_mesh_native(mesh, 8, False)
    t2 = time.perf_counter()
    domains = decomp.decompose_native(mesh, 8, None, npart, False, args.overlap)
    t3 = time.perf_counter()
    a = decomp.global_matrix(mesh)
    b = decomp.rhs_default(a)
    counts = np.bincount(npart, minlength=8)
    facts = {"generator_s": t1 - t0, "partition_s": t2 - t1, "decompose_s": t3 - t2, "partitioner": "csrc/partition.cpp "
             "(GeneoPartMeshNodal: multilevel recursive bisection, the METIS_PartMeshNodal stand-in)", "edge_cut": int(cut),
             "part_sizes": [int(c) for c in counts], "global_nnz": int(a.nnz)}
    desc = ("tst/graph generator (reference graph.cpp: --size %d --level 2 --noGround), %d nodes, nodal k-way partition into 8 "
            "(edge cut %d), overlap %d: irregular CSR" % (args.graph_size, mesh.nbNode, cut, args.overlap))
    return desc, mesh.nbNode, 8, domains, _WholePlan(mesh.nbNode), b, facts


def cpu_baseline(args, doms, lib):
    """The oracle's CPU kernels / algorithm on this box's host cores (rank 0, N = 1, bounded sample).  The GenEO leg
    also runs the GPU library on the SAME sample grid with the SAME options, so that the line carries both PCG
    iteration counts side by side (`parity_sample`)."""
    import scipy.sparse as sp
    out = {"kind": "port", "unit": "GB/s"}
    # (1) CSR SpMV, C + OpenMP restatement of MatMult_SeqAIJ, on the same block-diagonal local matrix
    so = os.path.join(ROOT, "oracle", "liboracle_kernels.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "csrc")])
    olib = C.CDLL(so)
    olib.oracle_num_threads.restype = C.c_int
    # bounded sample: the local matrices of as many subdomains as stay under 8 M rows (368^3: one 6.5 M-row subdomain)
    take, rows = [], 0
    for d in doms:
        if take and rows + d.a_dir.shape[0] > 8_000_000:
            break
        take.append(d)
        rows += d.a_dir.shape[0]
    a = sp.block_diag([d.a_dir for d in take], format="csr") if len(take) > 1 else take[0].a_dir.tocsr()
    rp, col, val = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)
    x = np.random.default_rng(0).random(a.shape[0])
    y = np.zeros(a.shape[0])
    # copies whose pages are first touched by the threads that read them (NUMA placement: arrays filled by this one
    # Python thread sit on one memory node of the two-socket host)
    olib.oracle_spmv_place.restype = C.c_void_p
    hp = C.c_void_p(olib.oracle_spmv_place(C.c_int(a.shape[0]), rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                           val.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)))
    olib.oracle_spmv_placed(hp)
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 5.0 and reps < 400:
        olib.oracle_spmv_placed(hp)
        reps += 1
    dt = (time.perf_counter() - t0) / max(1, reps)
    olib.oracle_spmv_result(hp, y.ctypes.data_as(C.c_void_p))
    olib.oracle_spmv_free(hp)
    nbytes = a.nnz * 12 + (a.shape[0] + 1) * 4 + a.shape[0] * 16
    out["value"] = nbytes / dt / 1e9
    out["cores"] = int(olib.oracle_num_threads())
    out["sample"] = "CSR SpMV of the %d-row / %d-nnz local matrix of %d of the rank's %d subdomains, %d repetitions (C + OpenMP)" % (
        a.shape[0], a.nnz, len(take), len(doms), reps)
    # (2) the SAME ALGORITHM compiled for the host cores: the library's own orchestration (LOBPCG with
    #     the AMG V-cycle, batched AMG-PCG local solves, E, PCG) over the OpenMP build of the test backend
    #     (tests/hostsim: -O3 -mavx2 -mfma -fopenmp), at a REAL size: ONE of the eight subdomains of BASELINE configs[1]
    #     (126^3 = 2.0 M DoF, 8 subdomains: a 67^3-class block of 286 k rows) with the bench's own options as a stand-alone
    #     problem -- its whole set-up and 25 PCG iterations on its local operator (the bounded sample of the contract; the
    #     whole 126^3 problem, GENEO_BENCH_CPU_WHOLE=1, took 118 + 48 s on the MI355X host).  Bench infrastructure: never a
    #     product backend.
    try:
        out["geneo_sample"] = cpu_geneo_sample(args)
    except Exception as e:
        out["geneo_sample"] = {"error": repr(e)}
    # (3) the oracle's GenEO setup + PCG solve (exact LU local solves, ARPACK shift-invert at -els2_eps_tol) on a small
    #     grid of the same workload and the GPU library beside it: the parity sample of the line
    try:
        from geneo4petsc_amd import decomp
        from geneo4petsc_amd.pc import GenEOPC
        from oracle import geneo_oracle as go
        ns = args.cpu_sample_n
        mesh = decomp.grid_mesh(n=ns, dim=3)
        dec = decomp.decompose(mesh, 8, None, decomp.structured_node_partition(ns, 3, (2, 2, 2)), False, args.overlap)
        am = decomp.global_matrix(mesh)
        bs = decomp.rhs_default(am)
        argv = geneo_argv(args)
        subs = [go.Subdomain(d.l2g, d.a_neu, d.mult, d.intersect) for d in dec.domains]
        t0 = time.perf_counter()
        orc = go.GenEOOracle(mesh.nbNode, subs, go.parse_options(argv))
        orc.dense_limit, orc.exact_eigs = 0, False      # the reference's literal call: ARPACK shift-invert at -els2_eps_tol
        # one worker process per subdomain (= one MPI rank of the reference each), at most the cores this process may use
        orc.workers = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        orc.setup(bs)
        t1 = time.perf_counter()
        res = go.solve(orc, bs, "cg", rtol=args.rtol)
        t2 = time.perf_counter()
        out["oracle_sample"] = {"grid": "%d^3 (%d DoF), 8 subdomains" % (ns, mesh.nbNode), "setup_s": t1 - t0,
                                "solve_s": t2 - t1, "iterations": res.its, "dimE": int(orc.dimE), "cores": int(orc.workers),
                                "note": "the oracle (exact LU + ARPACK): eigenproblems on %d worker processes (one per subdomain, "
                                        "as the reference's MPI ranks); local LU factorisations and the PCG loop on one core" % orc.workers}
        pc = GenEOPC(lib)
        pc.set_from_options(argv)
        pc.set_sizes(mesh.nbNode, 8)
        for d in dec.domains:
            pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
        pc.setup(bs)
        xg, gits, _, greason = pc.solve(bs)
        gi = pc.info()
        out["parity_sample"] = {"grid": "%d^3" % ns, "options": " ".join(argv),
                                "oracle_iterations": int(res.its), "gpu_iterations": int(gits),
                                "oracle_dimE": int(orc.dimE), "gpu_dimE": int(gi["dimE"]),
                                "oracle_kept": [int(v) for v in orc.realDimELoc],
                                "gpu_kept": [int(v) for v in pc.local_dims()],
                                "solution_rel_diff": float(np.linalg.norm(xg - res.x) / np.linalg.norm(res.x)),
                                "identical_counts": bool(int(res.its) == int(gits) and int(orc.dimE) == int(gi["dimE"])),
                                "gpu_setup_s": gi["setupTime"], "gpu_solve_s": gi["solveTime"]}
        pc.destroy()
    except Exception as e:     # the SpMV leg above is the contract; this leg is extra context
        out["oracle_sample"] = {"error": repr(e)}
    return out


def cpu_geneo_sample(args):
    """libgeneopc's host orchestration on the OpenMP build of the test backend, timed at 126^3 (see cpu_baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
    import build as hs_build
    from geneo4petsc_amd import _lib as L, decomp
    from geneo4petsc_amd.pc import GenEOPC
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # 64 threads: the row loops of a 286 k-row subdomain stop scaling there (MI355X host, 256 hardware threads: the whole
    # 126^3 problem took 118 + 48 s on all 256, OpenMP region overheads on the small coarse levels included)
    cores = max(1, min(64, avail))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    try:      # libgomp is already initialised (the SpMV leg): the environment is read only once
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(cores))
    except OSError:
        pass
    hlib = L.bind(hs_build.build_omp())
    assert hlib.GeneoBackendName() == b"host-openmp"
    n, parts = 126, (2, 2, 2)
    argv = geneo_argv(args)
    whole = bool(os.environ.get("GENEO_BENCH_CPU_WHOLE"))      # the whole 126^3 problem: minutes; default: one subdomain of it
    t0 = time.perf_counter()
    pc = GenEOPC(hlib)
    if whole:
        doms = [decomp.decompose_grid_domain(n, 3, parts, args.overlap, s, native=True) for s in range(8)]
        npart = decomp.structured_node_partition(n, 3, parts)
        b = np.zeros(n ** 3)
        for d in doms:
            rows = d.a_dir @ (d.l2g.astype(np.float64) + 1.0)
            sel = npart[d.l2g] == d.gid
            b[d.l2g[sel]] = rows[sel]
        pc.set_from_options(argv)
        pc.set_sizes(n ** 3, 8)
        for d in doms:
            pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
        what = "the whole %d^3 = %d DoF problem in 8 subdomains: full set-up + PCG solve" % (n, n ** 3)
    else:
        d = decomp.decompose_grid_domain(n, 3, parts, args.overlap, 0, native=True)
        nloc = len(d.l2g)
        b = d.a_neu @ (np.arange(nloc, dtype=np.float64) + 1.0)
        pc.set_from_options(argv + ["-ksp_max_it", "25", "-ksp_rtol", "1e-30", "-ksp_atol", "1e-300"])
        pc.set_sizes(nloc, 1)
        pc.add_subdomain(0, np.arange(nloc), d.mult, d.a_neu, d.a_dir)
        what = ("ONE of the eight subdomains of the %d^3 decomposition (%d local rows) as a stand-alone problem: its whole "
                "set-up (hierarchies, LOBPCG, Z) + 25 PCG iterations on its local operator" % (n, nloc))
    prep = time.perf_counter() - t0
    pc.setup(b)
    x, its, _, reason = pc.solve(b)
    info = pc.info()
    pc.destroy()
    return {"kind": "port", "what": what, "setup_s": info["setupTime"], "solve_s": info["solveTime"],
            "setup_plus_solve_s": info["setupTime"] + info["solveTime"], "iterations": int(its), "converged": reason,
            "dimE": int(info["dimE"]), "eig_iterations": int(info["eig_iterations"]), "cores": int(cores),
            "host_prep_s": prep, "options": " ".join(argv),
            "build": "tests/hostsim/backend_host.cpp + csrc/core.cpp, amg.cpp: g++ -O3 -mavx2 -mfma -fopenmp -DGENEO_HOST_OMP"}


def one_rank_of(args):
    """bench.py --one-rank-of 8: see the option's help.  The subdomain is handed to the library as a one-subdomain problem
    in its own numbering (N = n_loc, identity map) with the REAL multiplicities, A_Neu and A_Dir of the 8-subdomain
    decomposition: the pencil, both hierarchies, LOBPCG, Z and the local solves are exactly rank r's; the halo exchanges and
    all-reduces (none inside these phases) are simply absent.  The `solve` leg runs exactly --apply-count PCG
    iterations on (A_Neu, M^-1) -- one local SpMV + one preconditioner application + the dots each, the per-iteration work of
    the rank -- not a solve of the global system."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: the GenEO hot path has no CPU fallback")
    torch.cuda.set_device(0)
    from geneo4petsc_amd import _lib, decomp
    from geneo4petsc_amd.pc import GenEOPC, DeviceVector
    lib = _lib.load()
    lib.GeneoSetStream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    npg = args.n_per_gpu or 184
    n = args.n if args.n else 2 * npg
    parts = (2, 2, 2)
    t_prep = time.perf_counter()
    dom = decomp.decompose_grid_domain(n, 3, parts, args.overlap, args.rank_index, native=True)
    nloc = len(dom.l2g)
    b = dom.a_neu @ (np.arange(nloc, dtype=np.float64) + 1.0)
    prep_s = time.perf_counter() - t_prep
    argv = geneo_argv(args) + ["-ksp_max_it", str(args.apply_count), "-ksp_rtol", "1e-30", "-ksp_atol", "1e-300"]
    pc = GenEOPC(lib)
    pc.set_from_options(argv)
    pc.set_sizes(nloc, 1)
    pc.add_subdomain(0, np.arange(nloc), dom.mult, dom.a_neu, dom.a_dir)
    bd = DeviceVector.from_host(lib, b)

    def step():
        pc.setup(bd)
        x, its, rnorm, reason = pc.solve(bd)
        info = pc.info()
        x.free()
        return its, reason, info

    cold_t0 = time.perf_counter()
    first = step()                      # the first set-up of the process: every device block comes from hipMalloc
    torch.cuda.synchronize()
    cold = (time.perf_counter() - cold_t0, first[2]["setupTime"], first[2]["solveTime"])
    for _ in range(max(0, args.warmup - 1)):
        step()
    lib.GeneoKernelProfileStart(4, C.c_double(0.0))
    lib.GeneoDeviceMemInfo(None, None, None, None, None, None, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib.GeneoKernelProfileStop()
    mem = device_memory(lib)
    its, reason, info = last
    kernels = kernel_table(lib, 1e3 * elapsed, args.steps)
    ms_sum, by_sum = C.c_double(0), C.c_double(0)
    nsamp, nlaunch = C.c_longlong(0), C.c_longlong(0)
    lib.GeneoKernelProfileGet(0, C.byref(ms_sum), C.byref(by_sum), None, C.byref(nsamp), C.byref(nlaunch))
    gbs = by_sum.value / max(ms_sum.value, 1e-9) * 1e-6
    pc.setup(bd)                        # untimed step with the in-situ timer off (inner PCG chunks replay as HIP graphs)
    xg, gits, _, _ = pc.solve(bd)
    ginfo = pc.info()
    xg.free()
    dom_k = max(kernels, key=lambda k: k["share_of_step"]) if kernels else None
    roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
            "kernel": "k_spmv_sell"}
    if dom_k is not None:
        roof = {"bound": "hbm", "achieved": dom_k["hbm_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom_k["hbm_GBs"] / HBM_PEAK_GBS, "traffic": None, "kernel": dom_k["kernel"],
                "share_of_step": dom_k["share_of_step"], "avg_launch_ms": dom_k["avg_launch_ms"],
                "launches_timed": dom_k["launches_timed"], "launches_total": dom_k["launches_total"],
                "algorithmic_bytes_per_launch": dom_k["algorithmic_bytes_per_launch"]}
    roof["kernels"] = kernels
    roof["spmv_in_situ"] = {"GBs": gbs, "frac": gbs / HBM_PEAK_GBS,
                            "note": "working set %.0f MB > 256 MiB Infinity Cache: HBM-resident" % (by_sum.value / max(1, nsamp.value) / 1e6)}
    out = {
        "metric": "GenEO-PCG setup+solve sec and SpMV GB/s, 3D Laplacian 50M DoF, 1/2/4/8 GPUs",
        "value": gbs, "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "device_mem_peak_gb": mem["device_mem_peak_gb"], "device_memory": mem,
        "config": {"workload": "ONE RANK'S SHARE of the metric's configuration, alone on one GPU: subdomain %d of the 2x2x2 "
                               "decomposition of the %d^3 = %d DoF 7-pt Laplacian (reference tst/laplacian generator), overlap %d => "
                               "%d local rows, -geneo_lvl %s, -geneo_cut %d, tau %.2f, -els2_eps_tol %g; set-up = every "
                               "communication-free phase (SURVEY 8e), 'solve' = exactly %d PCG iterations on the local "
                               "operator (local SpMV + M^-1 + dots each), no halo exchange, no all-reduce"
                               % (args.rank_index, n, n ** 3, args.overlap, nloc, args.lvl, args.cut, args.tau, args.eps_tol,
                                  args.apply_count),
                   "grid": n, "dof": n ** 3, "subdomains": 8, "subdomains_per_gpu": 1, "rank_index": args.rank_index,
                   "local_rows": nloc, "local_nnz": int(dom.a_neu.nnz), "overlap": args.overlap, "transport": "none (one rank of 8)"},
        "setup_s": info["setupTime"], "solve_s": info["solveTime"], "setup_plus_solve_s": info["setupTime"] + info["solveTime"],
        "first_setup_s": {"setup": cold[1], "solve": cold[2], "wall": cold[0],
                          "note": "first set-up of the process: allocator empty, every block from hipMalloc"},
        "iterations": its, "converged": reason, "dimE": info["dimE"], "eig_iterations": info["eig_iterations"],
        "eig_coarse_iterations": info.get("eigCoarseIterations"),
        "local_solve_cg_iterations": info["dls1_iterations"], "local_solves": info["dls1_solves"],
        "amg_levels": info["amg_levels"], "amg_setup_s": info["amgSetupTime"], "host_prep_s": prep_s,
        "setup_breakdown_s": {"level1_upload_and_amg": info["lvl1SetupMinvTimeLoc"],
                              "eigensolve_lobpcg": info["lvl2SetupEigTimeLoc"], "coarse_operator_E": info["lvl2SetupETimeLoc"]},
        "untimed_step_with_hip_graphs_s": {"setup": ginfo["setupTime"], "solve": ginfo["solveTime"], "iterations": gits},
        "solve_breakdown_s": {"local_solves": info["lvl1ApplyMinvTimeLoc"], "coarse_Zt": info["lvl2ApplyZtTimeLoc"],
                              "coarse_Einv": info["lvl2ApplyEinvTimeLoc"]},
        "roofline": roof,
    }
    out["bench_wall_s"] = round(time.perf_counter() - T_PROCESS0, 1)
    print(json.dumps(out), flush=True)
    pc.destroy()


def one_gpu_anchor(args, lib, torch):
    from geneo4petsc_amd import decomp
    from geneo4petsc_amd.pc import GenEOPC, DeviceVector
    n = args.n_per_gpu or 184
    parts = (2, 2, 2)
    t0 = time.perf_counter()
    doms = [decomp.decompose_grid_domain(n, 3, parts, args.overlap, s, native=True) for s in range(8)]
    npart = decomp.structured_node_partition(n, 3, parts)
    b = np.zeros(n ** 3)
    for d in doms:
        rows = d.a_dir @ (d.l2g.astype(np.float64) + 1.0)
        sel = npart[d.l2g] == d.gid
        b[d.l2g[sel]] = rows[sel]
    prep = time.perf_counter() - t0
    pc = GenEOPC(lib)
    pc.set_from_options(geneo_argv(args))
    pc.set_sizes(n ** 3, 8)
    for d in doms:
        pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
    bd = DeviceVector.from_host(lib, b)
    res = None
    for _ in range(2):               # one warm-up, one measured step
        torch.cuda.synchronize()
        tw = time.perf_counter()
        pc.setup(bd)
        x, its, _, reason = pc.solve(bd)
        torch.cuda.synchronize()
        info = pc.info()
        x.free()
        res = {"workload": "%d^3 = %d DoF in 8 subdomains on ONE GPU (the per-GPU DoF count of the N > 1 runs)" % (n, n ** 3),
               "setup_s": info["setupTime"], "solve_s": info["solveTime"], "setup_plus_solve_s": info["setupTime"] + info["solveTime"],
               "wall_s": time.perf_counter() - tw, "iterations": int(its), "converged": reason, "dimE": info["dimE"],
               "host_prep_s": prep}
    pc.destroy()
    return res


def spawn(args):
    """N > 1 without torchrun: start the ranks as CHILD processes (this parent never touches the GPU) and relay."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # (torchrun's own parser would read a bare `--n` behind the script path as an abbreviation of its --nnodes / --nproc...)
    fwd = ["--grid" if a == "--n" else a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + fwd
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def device_memory(lib):
    """device-memory figures of the bench line (GiB): high-water mark of the library's live blocks over the timed steps,
    of live + parked in its caching allocator (the process's footprint on the card as the library sees it), and what
    hipMemGetInfo reports for the card right now (used = total - free: includes torch's context and the caches)"""
    v = [C.c_double(0) for _ in range(6)]
    lib.GeneoDeviceMemInfo(*[C.byref(x) for x in v], 0)
    g = 1073741824.0
    live, live_peak, foot_peak, cached, free, total = [x.value / g for x in v]
    return {"device_mem_peak_gb": live_peak, "device_mem_footprint_peak_gb": foot_peak, "device_mem_live_gb": live,
            "allocator_cache_gb": cached, "hip_mem_used_gb": total - free, "hip_mem_total_gb": total}


def kernel_table(lib, step_ms_total, n_steps):
    rows = []
    for cls, name, bound in KERNEL_CLASSES:
        ms, by, fl = C.c_double(0), C.c_double(0), C.c_double(0)
        ns, nl = C.c_longlong(0), C.c_longlong(0)
        lib.GeneoKernelProfileGet(cls, C.byref(ms), C.byref(by), C.byref(fl), C.byref(ns), C.byref(nl))
        if ns.value == 0:
            continue
        avg = ms.value / ns.value
        if bound == "hbm":
            ach, peak, unit = by.value / ms.value * 1e-6, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = fl.value / ms.value * 1e-9, FP64_MFMA_PEAK_TFS, "TFLOP/s"
        rows.append({"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                     "avg_launch_ms": avg, "launches_timed": int(ns.value), "launches_total": int(nl.value),
                     "algorithmic_bytes_per_launch": by.value / ns.value, "flops_per_launch": fl.value / ns.value,
                     "hbm_GBs": by.value / ms.value * 1e-6,
                     "share_of_step": avg * nl.value / max(step_ms_total, 1e-9)})
    return rows


def spmv_hbm_resident(lib, doms):
    """The fine-level SpMV with its working set forced out of the 256 MiB Infinity Cache: a 640 MB stream between
    launches (k_axpby on two 40 M-element vectors), HIP events around every SpMV."""
    import scipy.sparse as sp
    from geneo4petsc_amd.pc import Spmv, DeviceVector
    a = sp.block_diag([d.a_dir for d in doms], format="csr") if len(doms) > 1 else doms[0].a_dir.tocsr()
    h = Spmv(a, lib)
    x = DeviceVector.from_host(lib, np.random.default_rng(0).random(a.shape[0]))
    y = DeviceVector(lib, a.shape[0])
    big = 40_000_000
    u, v = DeviceVector.from_host(lib, np.ones(big)), DeviceVector.from_host(lib, np.ones(big))
    # the solver's fine matrix carries 16-bit column offsets (its single-precision companion): build them first, so that
    # the kernel timed here is the variant the local solves launch (10 B per entry)
    lib.GeneoSpmvFusedSingle(h.h, 0, x.ptr, y.ptr, None, None, None, C.c_double(0.0))
    lib.GeneoKernelProfileStart(1, C.c_double(1.0))
    for _ in range(12):
        lib.GeneoTestAxpby(u.ptr, v.ptr, C.c_double(0.5), C.c_double(0.5), big)
        lib.GeneoSpmvApply(h.h, x.ptr, y.ptr)
    lib.GeneoKernelProfileStop()
    ms, by, ns = C.c_double(0), C.c_double(0), C.c_longlong(0)
    lib.GeneoKernelProfileGet(0, C.byref(ms), C.byref(by), None, C.byref(ns), None)
    for t in (x, y, u, v):
        t.free()
    h.destroy()
    return {"GBs": by.value / max(ms.value, 1e-9) * 1e-6, "avg_launch_ms": ms.value / max(1, ns.value),
            "launches_timed": int(ns.value), "working_set_bytes": by.value / max(1, ns.value),
            "how": "640 MB evicting stream between launches"}


def main():
    args = build_parser().parse_args()
    if args.one_rank_of:
        return one_rank_of(args)
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, size))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: the GenEO hot path has no CPU fallback")
    staged = args.comm == "staged"       # rehearsal with ranks sharing a GPU: gloo + host-staged halo exchange
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
    # the library's own threads (side-stream set-up, upload helpers) are bound to this device, not to GPU 0
    if lib.GeneoSetDevice(local_rank) != local_rank:
        raise SystemExit("bench.py: rank %d could not select GPU %d" % (rank, local_rank))
    lib.GeneoSetStream(C.c_void_p(torch.cuda.current_stream().cuda_stream))

    t_prep = time.perf_counter()
    wl_desc, wl_facts = None, {}
    if args.workload != "laplacian":
        if size != 1:
            raise SystemExit("--workload %s runs on one GPU (N = 1)" % args.workload)
        wl_desc, ndof, nb, doms, plan, b, wl_facts = build_problem_unstructured(args)
        n, spg = int(round(ndof ** (1.0 / 3.0))), 8
    else:
        n, nb, spg, doms, plan, b = build_problem(args, rank, size)
        ndof = n ** 3
    comm = None
    comm_name = "none"
    transport_fallback = None      # set when the C++ RCCL transport could not start and torch.distributed carried the run
    if size > 1:
        from geneo4petsc_amd import comm as gcomm
        if staged:
            comm, comm_name = gcomm.StagedComm(plan, lib), "staged (gloo, host copies)"
        elif args.comm == "rccl":
            try:
                comm, comm_name = gcomm.RcclComm(plan, lib, dist, torch.device("cuda", local_rank)), "rccl (C++ transport in libgeneopc)"
            except RuntimeError as e:      # raised on every rank together (comm.RcclComm agrees first): same fallback everywhere
                transport_fallback = str(e)
                if rank == 0:
                    print("bench.py: %s -- falling back to the torch.distributed transport" % e, file=sys.stderr, flush=True)
                comm, comm_name = gcomm.TorchComm(plan, torch.device("cuda", local_rank)), "torch.distributed (nccl backend; the C++ RCCL transport failed to start)"
        else:
            comm, comm_name = gcomm.TorchComm(plan, torch.device("cuda", local_rank)), "torch.distributed (nccl backend)"
    prep_s = time.perf_counter() - t_prep
    if args.workload == "heat" and args.lvl == "SRAS,1":
        args.lvl = "ASM,1"          # configs[3]: GenEO coarse space (ASM,1) against plain ASM (ASM,0)
    argv = geneo_argv(args)
    bd = DeviceVector.from_host(lib, b)

    # ONE preconditioner object, set up again for every step (PCSetUp_GenEO releases the previous set-up first): the
    # subdomain matrices are handed over once, as in the reference's initGenEOPC, and only one set-up is alive at a time
    pc = GenEOPC(lib)
    pc.set_from_options(argv)
    pc.set_sizes(ndof, nb)
    if comm is not None:
        comm.attach(pc)
    for d in doms:
        pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        pc.setup(bd)
        x, its, rnorm, reason = pc.solve(bd)
        info = pc.info()
        x.free()
        return its, reason, info

    first_setup = None
    for w in range(args.warmup):
        tw = time.perf_counter()
        fi = step()[2]
        if w == 0:      # the first set-up of the process: the allocator is empty, every device block comes from hipMalloc
            torch.cuda.synchronize()
            first_setup = {"setup": fi["setupTime"], "solve": fi["solveTime"], "wall": time.perf_counter() - tw,
                           "note": "first set-up of the process (allocator empty, hipMalloc included); the timed steps re-set-up "
                                   "the same PC on cached device blocks"}
    # in-situ kernel timer: every 4th launch of each hot kernel class (fine-level SpMV / SpMM, MFMA Gram and update)
    lib.GeneoKernelProfileStart(4, C.c_double(0.0))
    lib.GeneoDeviceMemInfo(None, None, None, None, None, None, 1)      # high-water marks of the timed steps alone
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.GeneoKernelProfileStop()
    mem = device_memory(lib)
    its, reason, info = last
    kernels = kernel_table(lib, 1e3 * elapsed, args.steps)
    ms_sum, by_sum = C.c_double(0), C.c_double(0)
    nsamp, nlaunch = C.c_longlong(0), C.c_longlong(0)
    lib.GeneoKernelProfileGet(0, C.byref(ms_sum), C.byref(by_sum), None, C.byref(nsamp), C.byref(nlaunch))
    # One more, UNTIMED step with the in-situ timer off: the inner PCG chunks then replay as HIP graphs (the timer
    # needs direct launches: HIP events cannot bracket kernels inside a graph), which is how the library runs
    # outside this benchmark.  Reported as information only.
    pc.setup(bd)
    xg, gits, _, greason = pc.solve(bd)
    ginfo = pc.info()
    # true residual || A x - b || / || b || of that solve (driver:1072-1087), owned rows, summed over the ranks
    axg = pc.matmult(xg)
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
    gbs_rank = by_sum.value / max(ms_sum.value, 1e-9) * 1e-6
    if rank == 0:
        info_rows = sum(len(d.l2g) for d in doms)
        spmv_ws = by_sum.value / max(1, nsamp.value)
        cache_note = ("working set %.0f MB > 256 MiB Infinity Cache: HBM-resident" % (spmv_ws / 1e6)
                      if spmv_ws > 256 * 2 ** 20 else
                      "working set %.0f MB < 256 MiB Infinity Cache: this in-situ rate is cache-assisted, the HBM-resident "
                      "rate of the same kernel and matrix is roofline.spmv_hbm_resident" % (spmv_ws / 1e6))
        # HBM traffic per launch from the PMC passes (FETCH_SIZE x 2 / WRITE_SIZE, separate passes, calibrated in the same
        # run as the guide prescribes: scripts/pmc.py, committed as profiles/r02_hbm_traffic_pmc.json).  The PMC workload
        # launches each kernel class on this matrix / these block shapes; the in-situ launches of a class mix variants
        # (epilogues, widths), so the measured RATIO traffic / algorithmic of the class is applied to the in-situ
        # algorithmic bytes per launch.
        # The file names the kernel source it was measured on (sha256 of csrc/backend_hip.hip): a profile older than the
        # kernels is NOT applied -- traffic stays null and the line says why.
        traffic_note = None
        try:
            import hashlib
            pmc_path = os.path.join(ROOT, "profiles", "r04_hbm_traffic_pmc.json")
            prof = json.load(open(pmc_path))
            sha = hashlib.sha256(open(os.path.join(ROOT, "geneo4petsc_amd", "csrc", "backend_hip.hip"), "rb").read()).hexdigest()[:16]
            if prof.get("kernel_source_sha16") != sha:
                traffic_note = ("profiles/r04_hbm_traffic_pmc.json was measured on kernel source %s, this is %s: stale, not applied"
                                % (prof.get("kernel_source_sha16"), sha))
                print("bench.py: " + traffic_note, file=sys.stderr, flush=True)
                prof = {}
            for k in kernels:
                key = k["kernel"].split(" ")[0]
                hits = [v for name, v in prof.items() if isinstance(v, dict) and (name == key or name.startswith(key + "<") or name.startswith(key + "_ld"))
                        and "traffic_over_algorithmic" in v]
                hits.sort(key=lambda v: -v["traffic_over_algorithmic"])      # several forms of a class measured: the worst ratio
                # the ratio of a class is a property of the matrix the XCDs walk: applied when the profile was measured on
                # this rank's rows or on ONE of its (equal) subdomains
                prow = hits[0].get("rows", info_rows) if hits else 0
                if hits and (abs(prow - info_rows) <= 0.01 * info_rows or abs(prow - info_rows / max(1, len(doms))) <= 0.01 * prow):
                    k["traffic_over_algorithmic"] = hits[0]["traffic_over_algorithmic"]
                    k["traffic"] = hits[0]["traffic_over_algorithmic"] * k["algorithmic_bytes_per_launch"]
        except Exception as e:
            traffic_note = "no PMC profile applied: %r" % (e,)
        # MFMA utilisation of the Rayleigh-Ritz kernels (north_star: "MFMA utilisation for the Rayleigh-Ritz step"): PMC
        # passes of scripts/pmc.py (scripts/gpu.sh pmcm), attached to the two MFMA rows under the same staleness rule
        try:
            mp = json.load(open(os.path.join(ROOT, "profiles", "r04_mfma_pmc.json")))
            if mp.get("kernel_source_sha16") == sha:
                for k in kernels:
                    key = k["kernel"].split(" ")[0]
                    hits = {name: v for name, v in mp.items() if isinstance(v, dict) and name.startswith(key) and "MfmaUtil" in v}
                    if hits:
                        k["mfma_util_percent"] = {name: v["MfmaUtil"] for name, v in hits.items()}
                        k["mfma_util_source"] = "profiles/r04_mfma_pmc.json (rocprofv3 --pmc MfmaUtil, same kernel source)"
            elif rank == 0:
                print("bench.py: profiles/r04_mfma_pmc.json is older than the kernels: MFMA utilisation not attached", file=sys.stderr, flush=True)
        except Exception:
            pass
        dom = max(kernels, key=lambda k: k["share_of_step"]) if kernels else None
        roof = {"bound": "hbm", "achieved": gbs_rank, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_rank / HBM_PEAK_GBS,
                "traffic": None, "kernel": "k_spmv_sell"}
        if dom is not None:
            roof = {"bound": "hbm", "achieved": dom["hbm_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": dom["hbm_GBs"] / HBM_PEAK_GBS, "traffic": dom.get("traffic"),
                    "traffic_source": ("profiles/r04_hbm_traffic_pmc.json (same kernel source): PMC traffic / algorithmic of this "
                                       "kernel class (%.3f) x the in-situ algorithmic bytes per launch" % dom["traffic_over_algorithmic"])
                    if dom.get("traffic") else traffic_note,
                    "kernel": dom["kernel"], "share_of_step": dom["share_of_step"],
                    "avg_launch_ms": dom["avg_launch_ms"], "launches_timed": dom["launches_timed"],
                    "launches_total": dom["launches_total"],
                    "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                    "note": "dominant kernel = largest share of the step among the in-situ timed classes; its HBM "
                            "bandwidth is quoted against the 8 TB/s spec peak (MFMA kernels also list their TFLOP/s "
                            "fraction in kernels[]: they sit at the ridge)"}
        roof["kernels"] = kernels
        roof["spmv_in_situ"] = {"GBs": gbs_rank, "frac": gbs_rank / HBM_PEAK_GBS, "note": cache_note}
        out = {
            "metric": "GenEO-PCG setup+solve sec and SpMV GB/s, 3D Laplacian 50M DoF, 1/2/4/8 GPUs",
            "value": agg_gbs, "unit": "GB/s", "n_gpus": size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling if args.workload == "laplacian" else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "device_mem_peak_gb": mem["device_mem_peak_gb"], "device_memory": mem, "eig_groups": info.get("eigGroups"),
            "eig_coarse_iterations": info.get("eigCoarseIterations"),   # > 0: the eigensolves started from the multigrid level-1 pencil
            "transport_fallback": transport_fallback,      # not None: this is NOT a measurement of the C++ RCCL transport
            "config": {"workload": (wl_desc + ", -geneo_lvl %s, -geneo_cut %d, tau %.2f, -els2_eps_tol %g, PCG rtol %.0e"
                                    % (args.lvl, args.cut, args.tau, args.eps_tol, args.rtol)) if wl_desc else
                                   "3D 7-pt Laplacian (reference tst/laplacian generator, kappa=1, eps=1e-4; the reference has "
                                   "no 27-pt generator), %d^3 = %d DoF, %d subdomains (%d per GPU), overlap %d, "
                                   "-geneo_lvl %s, -geneo_cut %d, tau %.2f, -els2_eps_tol %g, PCG rtol %.0e; %s"
                                   % (n, n ** 3, nb, spg, args.overlap, args.lvl, args.cut, args.tau, args.eps_tol, args.rtol,
                                      ("the metric's configuration (BASELINE configs[2]): the whole %d^3 grid in 8 subdomains on %d GPU%s, "
                                       "eigensolves in %d group(s) under the device-memory budget" % (n, size, "" if size == 1 else "s", info.get("eigGroups") or 1))
                                      if args.scaling == "strong" else
                                      ("N=1: BASELINE configs[1] size (126^3 = 2.0 M DoF) in 8 subdomains on the one GPU"
                                       if size == 1 else
                                       "N>1: 184^3 DoF and one subdomain per GPU (N=8: 368^3 = 49.8 M), weak scaling")),
                       "workload_kind": args.workload, "workload_facts": wl_facts,
                       "local_rows": info_rows, "local_nnz": int(sum(d.a_neu.nnz for d in doms)),
                       "grid": n, "dof": ndof, "subdomains": nb, "subdomains_per_gpu": spg, "overlap": args.overlap,
                       "transport": comm_name,
                       "inner_solver": "local solves = AMG-PCG to -dls1_ksp_rtol %g in FP64 (the V-cycle streams float copies of "
                                       "its level matrices, -dls1_amg_precision single: FP64 arithmetic and vectors)" % args.dls1_rtol,
                       "allocator": "device blocks are cached across set-ups (the timed steps re-set-up one PC: no hipMalloc "
                                    "inside them after the warm-up step)"},
            "setup_s": setup_s, "solve_s": solve_s, "setup_plus_solve_s": setup_s + solve_s, "first_setup_s": first_setup,
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
            "roofline": roof,
        }
        # the oracle's counts at THIS grid, when a golden holds them (tests/golden/headline.json: 96^3, 126^3 at these options)
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "headline.json"))).get(str(n), {}).get("literal")
            default_opts = (args.workload == "laplacian" and nb == 8 and args.lvl == "SRAS,1" and args.cut == 20 and args.tau == 0.35
                            and args.eps_tol == 1e-3 and args.rtol == 1e-5 and args.overlap == 2 and not args.pc_args)
            if gold and default_opts:
                spread = sorted(set([gold["cg"]["its"]] + [c for v in gold.get("cg_spread", {}).values() for c in v]))
                out["golden_at_this_grid"] = {"source": "tests/golden/headline.json (reference-literal oracle: exact LU, ARPACK at 1e-3)",
                                              "oracle_pcg_iterations": gold["cg"]["its"], "oracle_pcg_spread_under_perturbation": spread,
                                              "oracle_dimE": gold["dimE"], "oracle_kept": gold["realDimELoc"],
                                              "gpu_pcg_iterations": int(its), "gpu_dimE": int(info["dimE"]),
                                              "dimE_identical": bool(gold["dimE"] == info["dimE"]),
                                              "pcg_count_identical": bool(gold["cg"]["its"] == int(its)),
                                              "pcg_count_inside_oracle_spread": bool(min(spread) <= int(its) <= max(spread))}
        except Exception:
            pass
        if args.workload == "heat":
            # configs[3]: the same problem with plain one-level ASM (-geneo_lvl ASM,0), PCG iteration counts side by side
            pc0 = GenEOPC(lib)
            pc0.set_from_options([a if a != args.lvl else "ASM,0" for a in argv] + ["-ksp_max_it", "5000"])
            pc0.set_sizes(ndof, nb)
            for d in doms:
                pc0.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
            pc0.setup(bd)
            x0v, its0, _, reason0 = pc0.solve(bd)
            i0 = pc0.info()
            x0v.free()
            pc0.destroy()
            out["asm0_vs_geneo"] = {"ASM,0": {"iterations": int(its0), "converged": reason0, "setup_s": i0["setupTime"], "solve_s": i0["solveTime"]},
                                    args.lvl: {"iterations": int(its), "converged": reason, "dimE": info["dimE"],
                                               "setup_s": setup_s, "solve_s": solve_s}}
        if size == 1 and not args.no_cpu_baseline and args.workload == "laplacian":
            if spmv_ws <= 256 * 2 ** 20:     # only where the in-situ rate is cache-assisted
                try:
                    roof["spmv_hbm_resident"] = spmv_hbm_resident(lib, doms)
                    roof["spmv_hbm_resident"]["frac"] = roof["spmv_hbm_resident"]["GBs"] / HBM_PEAK_GBS
                except Exception as e:
                    roof["spmv_hbm_resident"] = {"error": repr(e)}
            out["cpu_baseline"] = cpu_baseline(args, doms, lib)
            if "parity_sample" in out["cpu_baseline"]:
                out["parity_sample"] = out["cpu_baseline"].pop("parity_sample")
    pc.destroy()
    if rank == 0:
        if size > 1 and not args.no_anchor and args.scaling == "weak":
            # The weak-scaling curve compares N GPUs x (184^3, ONE subdomain each) with ... what on one GPU?  bench.py's N = 1
            # line is 126^3 in 8 subdomains (BASELINE configs[1]); the like-for-like anchor -- the SAME per-GPU DoF count on
            # one GPU, where the two-level method needs several subdomains -- is 184^3 in 8 subdomains, run here on rank 0's
            # GPU after the timed region (the other ranks wait at the barrier below).
            try:
                out["one_gpu_anchor"] = one_gpu_anchor(args, lib, torch)
            except Exception as e:
                out["one_gpu_anchor"] = {"error": repr(e)}
        out["bench_wall_s"] = round(time.perf_counter() - T_PROCESS0, 1)   # everything this process did: imports, host preparation, CPU legs, all steps
        print(json.dumps(out), flush=True)
    if comm is not None and hasattr(comm, "close"):
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
