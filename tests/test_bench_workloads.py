"""Host side of bench.py's workloads (no GPU): the heat / graph problems of BASELINE configs[3] / configs[4] in small, built
by the same functions the bench uses (C++ generator, partitioner and decomposition behind the C ABI), must be consistent
decompositions of their assembled operators: sum R^T A_Neu R = A, A_Dir = R A R^T, b = A (1..N)."""
import numpy as np
import pytest
import scipy.sparse as sp

import bench
from geneo4petsc_amd import decomp


def check_problem(ndof, doms, b, a):
    acc = sp.csr_matrix((ndof, ndof))
    for d in doms:
        r = sp.csr_matrix((np.ones(len(d.l2g)), (np.arange(len(d.l2g)), d.l2g)), shape=(len(d.l2g), ndof))
        acc = acc + r.T @ d.a_neu @ r
        assert abs(d.a_dir - a[d.l2g][:, d.l2g]).max() <= 1e-12 * abs(a).max()
        assert d.mult.min() >= 1
    assert abs(acc - a).max() <= 1e-12 * abs(a).max()
    np.testing.assert_allclose(b, a @ np.arange(1.0, ndof + 1.0), rtol=1e-10)       # high contrast: cancellation in the row sums


def test_heat_workload_is_a_consistent_decomposition():
    args = bench.build_parser().parse_args(["--workload", "heat", "--n", "12", "--kappa", "100"])
    desc, ndof, nb, doms, plan, b, facts = bench.build_problem_unstructured(args)
    assert ndof == 12 ** 3 and nb == 8 and len(doms) == 8 and "contrast 1e+06" in desc
    mesh = decomp.grid_mesh(n=12, dim=3, heat=True, lbd=1.0, dt=0.1, kappa_max=100.0, interp="minmax")
    a = decomp.global_matrix(mesh)
    assert a.data.max() / a.data[a.data > 0].min() > 1e5          # the contrast is in the operator
    check_problem(ndof, doms, b, a)


def test_graph_workload_is_a_consistent_decomposition():
    args = bench.build_parser().parse_args(["--workload", "graph", "--graph-size", "400"])
    desc, ndof, nb, doms, plan, b, facts = bench.build_problem_unstructured(args)
    mesh = decomp.graph_mesh(size=400, level=2, no_ground=True)
    assert ndof == mesh.nbNode == 9 * 20 * 20 and nb == 8 and len(doms) == 8
    assert sum(facts["part_sizes"]) == ndof and max(facts["part_sizes"]) - min(facts["part_sizes"]) <= 4 and facts["edge_cut"] > 0
    check_problem(ndof, doms, b, decomp.global_matrix(mesh))


def test_one_rank_of_8_flags_parse():
    args = bench.build_parser().parse_args(["--one-rank-of", "8", "--rank-index", "7", "--apply-count", "5"])
    assert args.one_rank_of == 8 and args.rank_index == 7 and args.apply_count == 5
    with pytest.raises(SystemExit):
        bench.build_parser().parse_args(["--one-rank-of", "4"])
