"""-m gpu: the N > 1 path of the HIP library (halo pack / unpack kernels, owner-halo layout, replicated coarse
solve, Krylov dots) with TWO ranks sharing the one GPU of the test box.  RCCL cannot put two ranks on one device,
so the transport is geneo4petsc_amd.comm.StagedComm (device buffers staged through the host, gloo between the
processes); everything numerical runs in libgeneopc.so on cuda:0.  Result = the serial oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
from oracle import geneo_oracle as go

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("lvl,ksp,parts,extra", [
    ("SRAS,1", "cg", (2, 2, 2), []),
    ("SORAS,2", "cg", (4, 2, 1), ["-geneo_tau", "0.02", "-geneo_gamma", "1.05", "-geneo_cut", "12", "-geneo_optim", "0.5"])])
def test_two_ranks_on_one_gpu_match_serial_oracle(tmp_path, lvl, ksp, parts, extra):
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", GENEO_WORKER_LIB="hip", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "tests", "gloo_worker.py"),
           out, lvl, ksp, ",".join(str(p) for p in parts)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    meta = json.loads(str(got["meta"]))
    mesh, dec, a, b = cases.grid_case(12, 3, parts, 1)
    argv = ["-geneo_lvl", lvl, "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", ksp, "-els2_eps_tol", "1e-10",
            "-ksp_rtol", "1e-6" if ksp == "cg" else "1e-8"] + extra     # cases.Tight: CG counts at 1e-6
    orc = cases.oracle_for(mesh, dec, argv, b)
    kspname, kw = cases.ksp_args(argv)
    res = go.solve(orc, b, kspname, **kw)
    assert meta["dims"] == orc.realDimELoc and meta["dimE"] == orc.dimE
    assert meta["reason"] == res.reason
    assert meta["its"] == res.its, (meta["its"], res.its)      # identical, CG included (no tolerance on the count)
    np.testing.assert_allclose(got["m"], orc.matmult(b), rtol=1e-12, atol=1e-9)
    assert np.linalg.norm(got["y"] - orc.apply(b)) <= 1e-9 * np.linalg.norm(orc.apply(b))
    # two iterates of the same count: to the Krylov tolerance for CG (cases.Tight: amplified rounding), 1e-7 for GMRES
    assert np.linalg.norm(got["x"] - res.x) <= (1e-6 if ksp == "cg" else 1e-7) * np.linalg.norm(res.x)
