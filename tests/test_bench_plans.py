"""The N > 1 layouts of bench.py -- --scaling strong (default: 8 subdomains of ONE grid on N = 2, 4, 8 ranks, 8 / N
subdomains per rank) and --scaling weak (one subdomain per rank) -- checked on the CPU: every rank's
ownership / halo plan is built the way the torchrun ranks build it (bench.build_problem, no GPU, no process group) and
the plans of all ranks are checked against each other -- what rank a sends to rank b is exactly, and in the same order,
what rank b expects from rank a (the RCCL transport posts ncclSend / ncclRecv pairs from these counts: a mismatch is a
hang on the 8-GPU node, not an error message)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


@pytest.mark.parametrize("size", [2, 4, 8])
def test_rank_plans_of_the_weak_scaling_layouts_agree(size):
    args = bench.build_parser().parse_args(["--gpus", str(size), "--scaling", "weak", "--n-per-gpu", "9", "--overlap", "2"])
    built = [bench.build_problem(args, r, size) for r in range(size)]
    n = built[0][0]
    assert n == int(round((9 ** 3 * size) ** (1.0 / 3.0)))
    plans = [b[4] for b in built]
    # every DOF has exactly one owner
    owned = np.concatenate([p.owned for p in plans])
    assert owned.size == n ** 3 and np.array_equal(np.sort(owned), np.arange(n ** 3))
    for a in range(size):
        pa = plans[a]
        assert len(built[a][3]) == 1                      # one subdomain per rank
        dom = built[a][3][0]
        # the rank's local space = owned + halo, and it is the subdomain's
        assert np.array_equal(np.sort(np.concatenate([pa.owned, pa.halo_gid])), np.sort(dom.l2g))
        assert pa.send_counts[a] == 0 and pa.recv_counts[a] == 0
        soff = np.concatenate([[0], np.cumsum(pa.send_counts)])
        for b in range(size):
            pb = plans[b]
            roff = np.concatenate([[0], np.cumsum(pb.recv_counts)])
            assert pa.send_counts[b] == pb.recv_counts[a], (a, b)
            sent = pa.owned[pa.send_idx[soff[b]:soff[b + 1]]]
            expected = pb.halo_gid[roff[a]:roff[a + 1]]
            assert np.array_equal(sent, expected), (a, b)
    # right-hand side b = A (1, 2, ..., N) on the owned rows: the pieces assemble the global vector
    from geneo4petsc_amd import decomp
    mesh = decomp.grid_mesh(n=n, dim=3)
    a_glob = decomp.global_matrix(mesh)
    b_ref = a_glob @ (np.arange(n ** 3, dtype=np.float64) + 1.0)
    for r in range(size):
        np.testing.assert_allclose(built[r][5], b_ref[plans[r].owned], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("size", [1, 2, 4, 8])
def test_rank_plans_of_the_strong_scaling_layouts_agree(size):
    """bench.py --scaling strong: the SAME grid in the SAME 8 subdomains (2 x 2 x 2 blocks) at every N, block (bi, bj, bk) on
    the rank whose box of the rank grid holds it -- 8 / 4 / 2 / 1 subdomains per rank.  Same checks as above, plus: the
    union of every rank's subdomains is the N = 1 decomposition (same l2g per global subdomain id)."""
    n = 14
    args = bench.build_parser().parse_args(["--gpus", str(size), "--n", str(n), "--overlap", "2"])
    assert args.scaling == "strong"
    built = [bench.build_problem(args, r, size) for r in range(size)]
    assert all(b[0] == n and b[1] == 8 and b[2] == 8 // size for b in built)
    plans = [b[4] for b in built]
    owned = np.concatenate([p.owned for p in plans])
    assert owned.size == n ** 3 and np.array_equal(np.sort(owned), np.arange(n ** 3))
    ref = bench.build_problem(bench.build_parser().parse_args(["--gpus", "1", "--n", str(n), "--overlap", "2"]), 0, 1)
    ref_l2g = {d.gid: d.l2g for d in ref[3]}
    seen = set()
    for a in range(size):
        pa = plans[a]
        doms = built[a][3]
        assert len(doms) == 8 // size
        for d in doms:
            assert d.gid not in seen and np.array_equal(d.l2g, ref_l2g[d.gid])
            seen.add(d.gid)
        touched = np.unique(np.concatenate([d.l2g for d in doms]))
        assert np.array_equal(np.sort(np.concatenate([pa.owned, pa.halo_gid])), touched)
        if size == 1:
            assert len(pa.halo_gid) == 0
            continue
        assert pa.send_counts[a] == 0 and pa.recv_counts[a] == 0
        soff = np.concatenate([[0], np.cumsum(pa.send_counts)])
        for b in range(size):
            pb = plans[b]
            roff = np.concatenate([[0], np.cumsum(pb.recv_counts)])
            assert pa.send_counts[b] == pb.recv_counts[a], (a, b)
            assert np.array_equal(pa.owned[pa.send_idx[soff[b]:soff[b + 1]]], pb.halo_gid[roff[a]:roff[a + 1]]), (a, b)
    assert seen == set(range(8))
    from geneo4petsc_amd import decomp
    mesh = decomp.grid_mesh(n=n, dim=3)
    b_ref = decomp.global_matrix(mesh) @ (np.arange(n ** 3, dtype=np.float64) + 1.0)
    for r in range(size):
        np.testing.assert_allclose(built[r][5], b_ref[plans[r].owned], rtol=1e-13, atol=1e-13)
