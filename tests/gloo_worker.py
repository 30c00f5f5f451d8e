"""Worker of tests/test_gloo.py: world_size ranks over gloo, each holding half of the subdomains,
running the multi-rank path of libgeneopc's host logic (test-only hostsim backend: "device"
pointers are host pointers, so the torch CPU staging buffers can be handed over directly)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def raw_exchange_check(lib, comm, plan, dist):
    """The two callbacks of the RCCL plan driven directly with two peers: forward (owners -> halos) and reverse (the
    count arrays swap roles), widths 1, 5 and 32, every value tagged with its sender and its position in the sender's
    buffer; the all-reduce; and the refusals (wider than the buffers, longer than the reduction buffer)."""
    import ctypes as C
    from geneo4petsc_amd.pc import GenEOPC
    pc = GenEOPC(lib)
    pc.set_sizes(12 ** 3, 2)
    comm.attach(pc)
    rank, size = plan.rank, plan.size
    counts = [None] * size
    dist.all_gather_object(counts, (np.asarray(plan.send_counts).tolist(), np.asarray(plan.recv_counts).tolist()))
    send, recv, red = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.GeneoRcclPlanBuffers(comm.h, 0, C.byref(send), C.byref(recv), C.byref(red)) == 0
    off = lambda c: np.concatenate([[0], np.cumsum(c)]).astype(int)
    for reverse in (0, 1):
        for w in (1, 5, 32):
            outc = plan.recv_counts if reverse else plan.send_counts
            inc = plan.send_counts if reverse else plan.recv_counts
            nout = int(np.sum(outc))
            src = (1000000.0 * (rank + 1) + np.arange(nout * w)).astype(np.float64)
            lib.GeneoH2D(send, src.ctypes.data_as(C.c_void_p), src.nbytes)
            assert lib.GeneoRcclPlanExchange(comm.h, 0, reverse | (w << 1)) == 0, lib.GeneoRcclGetError().decode()
            got = np.zeros(int(np.sum(inc)) * w)
            lib.GeneoD2H(got.ctypes.data_as(C.c_void_p), recv, got.nbytes)
            io = off(inc)
            for q in range(size):
                qout = counts[q][1] if reverse else counts[q][0]       # what q sends in this direction, per peer
                qo = off(qout)
                want = 1000000.0 * (q + 1) + np.arange(qo[rank] * w, qo[rank + 1] * w)
                np.testing.assert_array_equal(got[io[q] * w:io[q + 1] * w], want)
    vals = np.random.default_rng(rank).random(1000)
    allv = [None] * size
    dist.all_gather_object(allv, vals)
    lib.GeneoH2D(red, vals.ctypes.data_as(C.c_void_p), vals.nbytes)
    assert lib.GeneoRcclPlanAllreduce(comm.h, 0, 1000) == 0, lib.GeneoRcclGetError().decode()
    got = np.zeros(1000)
    lib.GeneoD2H(got.ctypes.data_as(C.c_void_p), red, got.nbytes)
    np.testing.assert_array_equal(got, sum(allv[1:], allv[0]))
    assert lib.GeneoRcclPlanExchange(comm.h, 0, 0 | (64 << 1)) != 0           # wider than the buffers: refused on every rank
    assert lib.GeneoRcclPlanAllreduce(comm.h, 0, (1 << 16) + 1) != 0           # longer than the reduction buffer: refused
    pc.destroy()


def main():
    out_path, lvl, ksp = sys.argv[1], sys.argv[2], sys.argv[3]
    parts = tuple(int(t) for t in sys.argv[4].split(",")) if len(sys.argv) > 4 else (2, 2, 2)
    extra = sys.argv[5:]
    use_hip = os.environ.get("GENEO_WORKER_LIB") == "hip"     # tests/test_gpu_multirank.py: both ranks on cuda:0
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    import hostsim_util as hu
    from geneo4petsc_amd import decomp
    from geneo4petsc_amd.comm import TorchComm, gather_owned
    from geneo4petsc_amd.pc import GenEOPC
    n, ov = 12, 1
    nb = parts[0] * parts[1] * parts[2]      # 8 in the mode tests; 2 = one subdomain per rank (the bench's N > 1 layout)
    sub_rank = np.arange(nb) * size // nb
    doms = [decomp.decompose_grid_domain(n, 3, parts, ov, s) for s in range(nb) if sub_rank[s] == rank]
    plan = decomp.grid_rank_plan(n, 3, parts, ov, sub_rank, rank, size, doms)
    if use_hip:
        from geneo4petsc_amd import _lib
        from geneo4petsc_amd.comm import StagedComm
        lib = _lib.load()
        comm = StagedComm(plan, lib)
    elif os.environ.get("GENEO_WORKER_LIB") == "staged":      # StagedComm logic itself, on the CPU backend
        from geneo4petsc_amd.comm import StagedComm
        lib = hu.hostsim_lib()
        comm = StagedComm(plan, lib)
    elif os.environ.get("GENEO_WORKER_LIB") == "rccl_standin":
        # the library's C++ RCCL transport (csrc/comm_rccl.cpp) with two peers: GENEO_RCCL_LIBRARY binds the test-only
        # shared-memory stand-in of librccl; gloo only carries the 128-byte unique id and the test's own gathers
        from geneo4petsc_amd.comm import RcclComm
        lib = hu.hostsim_lib()
        comm = RcclComm(plan, lib, dist, "cpu")
        raw_exchange_check(lib, comm, plan, dist)
    else:
        lib = hu.hostsim_lib()
        comm = TorchComm(plan, "cpu")
    pc = GenEOPC(lib)
    pc.set_from_options(["-geneo_lvl", lvl, "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", ksp,
                         "-els2_eps_tol", "1e-10", "-ksp_rtol", "1e-6" if ksp == "cg" else "1e-8"] + extra)
    pc.set_sizes(n ** 3, nb)
    comm.attach(pc)
    for d in doms:
        pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
        if lvl.endswith("2"):      # intersectLoc emptiness (GenEO-2 gamma_loc): required on several ranks
            pc.set_intersect(d.gid, [len(x) > 0 for x in d.intersect])
    # b = A (1..N) on the owned rows, from the Dirichlet rows of the domain that owns each node
    xstar = np.arange(1.0, n ** 3 + 1.0)
    b = np.zeros(len(plan.owned))
    boxes = decomp.grid_boxes(n, 3, parts)
    for d in doms:
        rows = d.a_dir @ xstar[d.l2g]
        sel = np.isin(d.l2g, plan.owned) & (decomp.structured_node_partition(n, 3, parts)[d.l2g] == d.gid)
        b[np.searchsorted(plan.owned, d.l2g[sel])] = rows[sel]
    pc.setup(b)
    x, its, rnorm, reason = pc.solve(b)
    y = pc.apply(b)
    m = pc.matmult(b)
    res = dict(its=its, reason=reason, dims=[int(v) for v in pc.local_dims()], dimE=pc.info()["dimE"],
               gamma=[float(v) for v in pc.local_params()[1]])
    xf = gather_owned(x, plan, n ** 3)
    yf = gather_owned(y, plan, n ** 3)
    mf = gather_owned(m, plan, n ** 3)
    bf = gather_owned(b, plan, n ** 3)
    if rank == 0:
        np.savez(out_path, x=xf, y=yf, m=mf, b=bf, meta=json.dumps(res))
    if comm.error is not None:
        raise comm.error
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
