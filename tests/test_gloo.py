"""N > 1 path on CPU: world_size 2 over gloo.  Each rank holds 4 of the 8 subdomains; halo exchange
and all-reduces go through geneo4petsc_amd.comm.TorchComm; the result must equal the serial oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
from oracle import geneo_oracle as go

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """a free rendezvous port per test (a fixed one collides with concurrent jobs on the same host)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("lvl,ksp,parts,extra", [
    ("ASM,1", "cg", (2, 2, 2), []), ("RAS,H1", "gmres", (2, 2, 2), []),
    ("SRAS,1", "cg", (2, 1, 1), []),        # ONE subdomain per rank: the layout of bench.py --gpus 8 (config 3)
    # bench.py --scaling strong at N = 2: four of the eight subdomains per rank, eigensolved group by group under the
    # device-memory budget (one subdomain per group here), the next group prepared on a helper thread
    ("SRAS,1", "cg", (2, 2, 2), ["-geneo_eig_group_rows", "1"]),
    ("SORAS,2", "cg", (4, 2, 1), ["-geneo_tau", "0.02", "-geneo_gamma", "1.05", "-geneo_cut", "12", "-geneo_optim", "0.5"])])
def test_two_ranks_match_serial_oracle(tmp_path, lvl, ksp, parts, extra):
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    if lvl == "RAS,H1":
        env["GENEO_WORKER_LIB"] = "staged"     # same path through the host-staged transport (comm.StagedComm)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "gloo_worker.py"),
           out, lvl, ksp, ",".join(str(p) for p in parts)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    meta = json.loads(str(got["meta"]))
    mesh, dec, a, b = cases.grid_case(12, 3, parts, 1)
    np.testing.assert_allclose(got["b"], b, rtol=1e-13)
    argv = ["-geneo_lvl", lvl, "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", ksp, "-els2_eps_tol", "1e-10",
            "-ksp_rtol", "1e-6" if ksp == "cg" else "1e-8"] + \
        [e for i, e in enumerate(extra) if "-geneo_eig_group_rows" not in (e, extra[i - 1] if i else "")]   # cases.Tight: CG counts at 1e-6
    orc = cases.oracle_for(mesh, dec, argv, b)
    kspname, kw = cases.ksp_args(argv)
    res = go.solve(orc, b, kspname, **kw)
    assert meta["dims"] == orc.realDimELoc and meta["dimE"] == orc.dimE
    if lvl.endswith("2"):      # rank 0 holds subdomains 0..3: gamma_loc from the all-reduced connectivity matrix
        np.testing.assert_allclose(meta["gamma"], orc.gammaLoc[:len(meta["gamma"])], rtol=1e-12)
    assert meta["reason"] == res.reason
    assert meta["its"] == res.its, (meta["its"], res.its)      # identical, CG included (no tolerance on the count)
    np.testing.assert_allclose(got["m"], orc.matmult(b), rtol=1e-12, atol=1e-9)
    assert np.linalg.norm(got["y"] - orc.apply(b)) <= 1e-9 * np.linalg.norm(orc.apply(b))
    # two iterates of the same count: to the Krylov tolerance for CG (cases.Tight: amplified rounding), 1e-7 for GMRES
    assert np.linalg.norm(got["x"] - res.x) <= (1e-6 if ksp == "cg" else 1e-7) * np.linalg.norm(res.x)


def test_rccl_bootstrap_failure_is_collective():
    """comm.RcclComm agrees among the ranks before raising (unique id on rank 0, then communicator creation): on this
    GPU-less box the C++ transport cannot start, and both ranks must learn that together instead of one of them hanging
    in the broadcast."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "rccl_bootstrap_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OUTCOME " in r.stdout
