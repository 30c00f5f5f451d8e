"""The REAL reference generators (tst/laplacian, tst/heat, tst/graph getInput plugins compiled from
/root/reference into oracle/_ref by oracle/ref_build/Makefile) against the oracle's restatement AND the
product's vectorised generators: same elements, same order, same element matrices."""
import numpy as np
import pytest

from geneo4petsc_amd import decomp
from oracle import driver_oracle as drv
from oracle import ref_generators as rg

pytestmark = pytest.mark.skipif(not rg.available(), reason="oracle/_ref not built (needs /root/reference)")

GRID = [("laplacian", "--size 5 --dim 3 --kappa 2. lin", dict(size=5, dim=3, kappa_max=2.0, interp="lin")),
        ("laplacian", "--size 7 --dim 2 --weakScaling 2", dict(size=7, dim=2, weak=2)),
        ("laplacian", "--size 6 --dim 1 --inpEps 0.01 --kappa 5 quad",
         dict(size=6, dim=1, inp_eps=0.01, kappa_max=5.0, interp="quad")),
        ("laplacian", "--size 10 --dim 3 --kappa 2. lin", dict(size=10, dim=3, kappa_max=2.0, interp="lin")),
        ("heat", "--size 4 --dim 3 --kappa 100 minmax --lbd 2. --dt 0.5",
         dict(size=4, dim=3, kappa_max=100.0, interp="minmax", heat=True, lbd=2.0, dt=0.5)),
        ("heat", "--size 5 --dim 3 --kappa 2. quad --lbd 1. --dt 0.1",
         dict(size=5, dim=3, kappa_max=2.0, interp="quad", heat=True, lbd=1.0, dt=0.1)),
        ("heat", "--size 6 --dim 2", dict(size=6, dim=2, heat=True))]
GRAPH = [("--size 9 --level 2", dict(size=9, level=2)),
         ("--size 16 --level 1 --noGround", dict(size=16, level=1, no_ground=True)),
         ("--size 10 --level 2 --noGround", dict(size=10, level=2, no_ground=True)),
         ("--size 4 --level 3 --weakScaling 4 --inpEps 0.1", dict(size=4, level=3, weak=4, inp_eps=0.1))]


def _same_as_oracle(ref, orc):
    assert (ref.nbElem, ref.nbNode) == (orc.nbElem, orc.nbNode)
    assert ref.elemPtr[:ref.nbElem + 1] == orc.elemPtr[:orc.nbElem + 1] and ref.elemIdx == orc.elemIdx
    for a, b in zip(ref.elemSubMat, orc.elemSubMat):
        np.testing.assert_allclose(a, b, rtol=1e-15, atol=0)


def _same_as_product(ref, pm):
    assert (ref.nbElem, ref.nbNode) == (pm.nbElem, pm.nbNode)
    for e in range(ref.nbElem):
        s, t = ref.elemPtr[e], ref.elemPtr[e + 1]
        nn = t - s
        assert list(pm.nodes[e][:nn]) == ref.elemIdx[s:t]
        np.testing.assert_allclose(pm.mats[e].reshape(2, 2)[:nn, :nn].ravel(), ref.elemSubMat[e], rtol=1e-15)


@pytest.mark.parametrize("plugin,args,kw", GRID, ids=[g[1] for g in GRID])
def test_grid_generators(plugin, args, kw):
    ref = rg.get_input(plugin, args)
    _same_as_oracle(ref, drv.grid_input(**kw))
    _same_as_product(ref, decomp.grid_mesh(**kw))


@pytest.mark.parametrize("args,kw", GRAPH, ids=[g[0] for g in GRAPH])
def test_graph_generator(args, kw):
    ref = rg.get_input("graph", args.replace(" ", "#"))     # '#' separators as on the reference CLI
    _same_as_oracle(ref, drv.graph_input(**kw))
    _same_as_product(ref, decomp.graph_mesh(**kw))
