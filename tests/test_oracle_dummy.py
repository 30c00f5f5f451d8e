"""Pin the oracle against every golden the reference holds for this path: tst/dummy/*.ref (84 logs).

What the goldens pin: per-rank MATIS local matrices (decomposition,
overlap, element-multiplicity weighting, local ordering), nnz count, RHS, converged solution
to PETSc print precision, PC name string.  They do NOT pin eigenvalues / dimE / iteration
counts (--shortRes) -- those are covered by tests/test_oracle_eig.py against dense ground truth.
"""
import numpy as np
import pytest

import dummy_cases as dc
from oracle import driver_oracle as drv
from oracle import geneo_oracle as go


def _build(rec):
    mesh = dc.mesh_for(rec)
    part = dc.partition_for(rec)
    assert part is not None, "no 2-way partition reproduces the golden local matrices: " + rec["file"]
    ep, npart = part
    dual = rec["metis"] == "dual"
    dec = drv.decompose(mesh, 2, list(ep) if ep else None, list(npart) if npart else None, dual,
                        rec["overlap"])
    aneu = [drv.assemble_local(mesh, dec, p) for p in range(2)]
    a = drv.global_matrix(dec, aneu, mesh.nbNode)
    return mesh, dec, aneu, a


@pytest.mark.parametrize("rec", dc.load()["refs"], ids=lambda r: r["file"][:-4])
def test_dummy_golden(rec):
    mesh, dec, aneu, a = _build(rec)
    b = dc.rhs_for(rec, a)
    nnz = sum(drv.local_nnz(m) for m in aneu)
    info = rec["info"]
    assert info[0].startswith("INFO: nb DOFs %d, nb elements %d, nnz coefs %d, nb partitions 2, overlap %d, metis %s"
                              % (mesh.nbNode, mesh.nbElem, nnz, rec["overlap"], rec["metis"]))
    np.testing.assert_allclose(b, rec["b"], rtol=1e-6)
    if rec["mat_type"] == "mpiaij":           # -pc_type bjacobi: assembled global matrix is printed
        assert dc.same_rows(dc.rows_of(a), rec["mats"][0])
        return
    for p in range(2):                         # MATIS local (Neumann) matrices, rank order
        assert dc.same_rows(dc.rows_of(aneu[p]), rec["mats"][p])
    args = ["-geneo_lvl", rec["geneo_lvl"]]
    if rec["geneo_cut"] > 0:
        args += ["-geneo_cut", str(rec["geneo_cut"])]
    if rec["offload"]:
        args += ["-geneo_offload"]
    o = go.parse_options(args)
    assert ("INFO: %s pc" % o.name) in info[2]
    subs = [go.Subdomain(dec.nodeIdxDom[p], aneu[p], dec.nodeIdxMult[dec.nodeIdxDom[p]],
                         dec.intersectDom[p]) for p in range(2)]
    orc = go.GenEOOracle(mesh.nbNode, subs, o).setup(b)
    res = go.solve(orc, b, "gmres", rtol=rec["ksp_rtol"], atol=rec["ksp_atol"])
    assert res.reason.startswith("KSP_CONVERGED")          # "INFO: solve - converged"
    np.testing.assert_allclose(res.x, rec["x"], rtol=1e-5, atol=1e-6)


def test_known_answer_tridiag_dual():
    """SURVEY.md section 8c known answer: tridiag, dual partition, GenEO-1 ASM."""
    rec = [r for r in dc.geneo_refs() if r["file"] == "tridiag-pc=geneoASM1-metis=dual.ref"][0]
    mesh, dec, aneu, a = _build(rec)
    b = dc.rhs_for(rec, a)
    assert list(dec.nodeIdxDom[0]) == [3, 4, 5, 6, 7] and list(dec.nodeIdxDom[1]) == [0, 1, 2, 3]
    np.testing.assert_allclose(b, [2, 4, 6, 8, 10, 12, 14, 25])
    subs = [go.Subdomain(dec.nodeIdxDom[p], aneu[p], dec.nodeIdxMult[dec.nodeIdxDom[p]],
                         dec.intersectDom[p]) for p in range(2)]
    orc = go.GenEOOracle(8, subs, go.parse_options(["-geneo_lvl", "ASM,1"])).setup(b)
    assert orc.nicolaidesLoc == [1, 1] and orc.realDimELoc == [1, 1]
    np.testing.assert_allclose(orc.E, np.diag([10.0, 8.0]), atol=1e-12)
    np.testing.assert_allclose(orc.apply_q(b), [2, 2, 2, 4.25, 6.5, 6.5, 6.5, 6.5], atol=1e-12)
    # generalized eigenvalues of (A_Neu, D A_Dir D), SURVEY 8c
    import scipy.linalg as sla
    for p, want in ((0, [0.928205128205, 1, 1, 1, 2]), (1, [0.928229665072, 1, 1, 2])):
        d = np.diag(orc.D[p])
        w = sla.eigvalsh(aneu[p].toarray(), d @ orc.a_dir[p].toarray() @ d)
        np.testing.assert_allclose(np.sort(w), want, rtol=1e-10)
