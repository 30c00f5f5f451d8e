"""The library's C++ RCCL transport (csrc/comm_rccl.cpp) with TWO peers, on CPU.

The GPU box has one MI355X and RCCL refuses two ranks per device, so the transport's N > 1 arithmetic -- per-peer
offsets, the forward / reverse role swap of the count arrays, widths up to 32 (blocked assembly of E), the capacity of
the reduction buffer -- cannot run there with more than one rank.  Here two processes on the test-only host backend
drive the real comm_rccl.cpp through PCGenEOSetCommRccl; its load_api() binds, through the GENEO_RCCL_LIBRARY path
override, a test-only stand-in of librccl (tests/rccl_standin: ncclSend / ncclRecv / ncclGroup* / ncclAllReduce between
processes over shared memory, host pointers).  gloo carries only the 128-byte unique id and the test's own gathers.
Asserted: the raw exchanges (tests/gloo_worker.py::raw_exchange_check) and a full GenEO set-up + solve equal to the
serial oracle, as tests/test_gloo.py does for comm.TorchComm.  Says nothing about xGMI: no hardware N > 1 run exists."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cases
from oracle import geneo_oracle as go

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_standin():
    """tests/rccl_standin/build_standin.py, loaded by path (tests/hostsim has a build.py of its own on sys.path)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("geneo_rccl_standin_build", os.path.join(ROOT, "tests", "rccl_standin", "build_standin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("lvl,ksp,parts,nproc", [
    ("SRAS,1", "cg", (2, 1, 1), 2),        # one subdomain per rank: bench.py's N = 2 layout
    ("RAS,H1", "gmres", (2, 2, 2), 2),     # four per rank, hybrid: every operator
    ("SRAS,1", "cg", (2, 2, 1), 4)])       # bench.py's N = 4 layout: every rank has three peers in one ncclGroup
def test_cpp_rccl_transport_two_peers_matches_serial_oracle(tmp_path, lvl, ksp, parts, nproc):
    standin = build_standin()
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", GENEO_WORKER_LIB="rccl_standin",
               GENEO_RCCL_LIBRARY=standin)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "gloo_worker.py"), out, lvl, ksp,
           ",".join(str(p) for p in parts)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    meta = json.loads(str(got["meta"]))
    mesh, dec, a, b = cases.grid_case(12, 3, parts, 1)
    np.testing.assert_allclose(got["b"], b, rtol=1e-13)
    argv = ["-geneo_lvl", lvl, "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", ksp, "-els2_eps_tol", "1e-10",
            "-ksp_rtol", "1e-6" if ksp == "cg" else "1e-8"]
    orc = cases.oracle_for(mesh, dec, argv, b)
    kspname, kw = cases.ksp_args(argv)
    res = go.solve(orc, b, kspname, **kw)
    assert meta["dims"] == orc.realDimELoc and meta["dimE"] == orc.dimE
    assert meta["reason"] == res.reason
    assert meta["its"] == res.its, (meta["its"], res.its)
    np.testing.assert_allclose(got["m"], orc.matmult(b), rtol=1e-12, atol=1e-9)
    assert np.linalg.norm(got["y"] - orc.apply(b)) <= 1e-9 * np.linalg.norm(orc.apply(b))
    assert np.linalg.norm(got["x"] - res.x) <= (1e-6 if ksp == "cg" else 1e-7) * np.linalg.norm(res.x)
