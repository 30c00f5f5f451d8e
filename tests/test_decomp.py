"""Product host code (geneo4petsc_amd/decomp.py, vectorised) against the loop-for-loop oracle
restatement of the reference driver and against the reference's tst/dummy goldens."""
import numpy as np
import pytest
import scipy.sparse as sp

import dummy_cases as dc
from geneo4petsc_amd import decomp
from oracle import driver_oracle as drv


def _same(a, b, tol=1e-13):
    d = (a - b)
    return a.shape == b.shape and (abs(d).max() if d.nnz else 0.0) <= tol


@pytest.mark.parametrize("dim,n,parts,ov,kw", [
    (3, 6, (2, 2, 1), 1, dict(kappa_max=2.0, interp="lin")),
    (2, 9, (3, 2, 1), 2, dict()),
    (3, 5, (2, 1, 2), 0, dict(heat=True, kappa_max=3.0, interp="minmax")),
    (1, 12, (3, 1, 1), 1, dict(kappa_max=2.0, interp="quad")),
    (3, 7, (2, 2, 2), 2, dict(heat=True, lbd=2.0, dt=0.5)),
])
def test_grid_generators_and_decomposition(dim, n, parts, ov, kw):
    om = drv.grid_input(size=n, dim=dim, **kw)
    pm = decomp.grid_mesh(size=n, dim=dim, **kw)
    assert (om.nbElem, om.nbNode) == (pm.nbElem, pm.nbNode)
    for e in range(om.nbElem):                       # same elements, same ORDER, same matrices
        s, t = om.elemPtr[e], om.elemPtr[e + 1]
        assert list(pm.nodes[e][:t - s]) == om.elemIdx[s:t]
        nn = t - s
        np.testing.assert_allclose(pm.mats[e].reshape(2, 2)[:nn, :nn].ravel(), om.elemSubMat[e], rtol=1e-15)
    nb = parts[0] * parts[1] * parts[2]
    npart = drv.structured_node_partition(drv.grid_size(n, 1, dim), dim, parts)
    assert (npart == decomp.structured_node_partition(drv.grid_size(n, 1, dim), dim, parts)).all()
    od = drv.decompose(om, nb, None, list(npart), False, ov)
    pd = decomp.decompose(pm, nb, None, npart, False, ov)
    a = drv.global_matrix(od, [drv.assemble_local(om, od, p) for p in range(nb)], om.nbNode)
    assert _same(a, decomp.global_matrix(pm))
    assert (od.nodeIdxMult == pd.node_mult).all() and (od.elemIdxMult == pd.elem_mult).all()
    for p in range(nb):
        assert (od.nodeIdxDom[p] == pd.domains[p].l2g).all()
        assert (od.nodeIdxMult[od.nodeIdxDom[p]] == pd.domains[p].mult).all()
        assert _same(drv.assemble_local(om, od, p), pd.domains[p].a_neu)
        l = od.nodeIdxDom[p]
        assert _same(a[l][:, l].tocsr(), pd.domains[p].a_dir)
        for q in range(nb):
            assert (od.intersectDom[p][q] == pd.domains[p].intersect[q]).all()


@pytest.mark.parametrize("kw", [dict(size=9, level=2), dict(size=16, level=1, no_ground=True),
                                dict(size=4, level=3, no_ground=True)])
def test_graph_generator(kw):
    om, pm = drv.graph_input(**kw), decomp.graph_mesh(**kw)
    assert (om.nbElem, om.nbNode) == (pm.nbElem, pm.nbNode)
    for e in range(om.nbElem):
        assert list(pm.nodes[e]) == om.elemIdx[2 * e:2 * e + 2]
        np.testing.assert_allclose(pm.mats[e], om.elemSubMat[e], rtol=1e-15)


def test_dual_partition_matches_oracle():
    om = drv.grid_input(size=5, dim=2)
    pm = decomp.grid_mesh(size=5, dim=2)
    rng = np.random.default_rng(0)
    ep = rng.integers(0, 3, size=om.nbElem)
    od = drv.decompose(om, 3, list(ep), None, True, 1)
    pd = decomp.decompose(pm, 3, ep, None, True, 1)
    for p in range(3):
        assert (od.nodeIdxDom[p] == pd.domains[p].l2g).all()
        assert _same(drv.assemble_local(om, od, p), pd.domains[p].a_neu)


@pytest.mark.parametrize("rec", dc.geneo_refs()[::7], ids=lambda r: r["file"][:-4])
def test_dummy_goldens_through_product_decomposition(rec):
    """Reference goldens: the product's reader + decomposition + weighted assembly reproduce the printed
    per-rank MATIS matrices, nnz count and RHS."""
    d = dc.load()
    mesh = decomp.read_input_text(d["inputs"][rec["input"] + ".inp"], rec["inpEps"])
    ep, npart = dc.partition_for(rec)
    dec = decomp.decompose(mesh, 2, ep, npart, rec["metis"] == "dual", rec["overlap"])
    for p in range(2):
        assert dc.same_rows(dc.rows_of(dec.domains[p].a_neu), rec["mats"][p])
    nnz = sum(int(dom.a_neu.indptr[-1]) for dom in dec.domains)
    assert ("nnz coefs %d," % nnz) in rec["info"][0]
    a = decomp.global_matrix(mesh)
    b = decomp.read_b_text(d["inputs"]["B.inp"], mesh.nbNode) if rec["use_b_file"] else decomp.rhs_default(a)
    np.testing.assert_allclose(b, rec["b"], rtol=1e-6)


def test_rank_plans_cover_every_halo():
    mesh = decomp.grid_mesh(n=8, dim=3)
    npart = decomp.structured_node_partition(8, 3, (2, 2, 1))
    dec = decomp.decompose(mesh, 4, None, npart, False, 1, build=False)
    sub_rank = np.array([0, 0, 1, 1])
    owner = sub_rank[npart]
    plans = decomp.rank_plans(dec, owner, sub_rank, 2)
    for pl in plans:
        assert (owner[pl.owned] == pl.rank).all() and (owner[pl.halo_gid] != pl.rank).all()
        assert pl.recv_counts.sum() == pl.halo_gid.size and pl.send_counts.sum() == pl.send_idx.size
    # what rank 0 sends to rank 1 is exactly what rank 1 expects from rank 0, in the same order
    s01 = plans[0].owned[plans[0].send_idx[plans[0].send_counts[:1].sum():plans[0].send_counts[:2].sum()]]
    r10 = plans[1].halo_gid[:plans[1].recv_counts[0]]
    assert (s01 == r10).all()


@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("n,k", [(10, 8), (9, 5)])
def test_kway_partitioner(n, k, dual):
    """Stand-in for METIS_PartMeshDual / Nodal (driver:381-445): balanced within 5 %, connected parts,
    deterministic, and -- with the multilevel spectral bisection of round 2 -- the cut of the structured 2x2x2 blocks
    on a cube, i.e. the optimum (round 1: within 2.5x)."""
    from scipy.sparse.csgraph import connected_components
    mesh = decomp.grid_mesh(size=n, dim=3)
    g = decomp.mesh_graph(mesh, dual)
    assert g.shape[0] == (mesh.nbElem if dual else mesh.nbNode) and (g != g.T).nnz == 0
    p = decomp.partition_graph(g, k)
    sizes = np.bincount(p, minlength=k)
    assert sizes.min() >= 1 and sizes.max() - sizes.min() <= max(1, 0.05 * g.shape[0] / k * 2)
    for q in range(k):
        assert connected_components(g[p == q][:, p == q])[0] == 1
    assert np.array_equal(p, decomp.partition_graph(g, k))
    if not dual and k == 8:
        blocks = decomp.structured_node_partition(n, 3, (2, 2, 2))
        assert decomp.edge_cut(g, p) <= 1.05 * decomp.edge_cut(g, blocks)
    ep, npart = decomp.partition_mesh(mesh, k, dual)
    dec = decomp.decompose(mesh, k, ep, npart, dual, 1)
    assert len(dec.domains) == k and all(len(d.l2g) > 0 for d in dec.domains)
    a = decomp.global_matrix(mesh)                        # the MATIS of the decomposition assembles back to A
    acc = sp.csr_matrix(a.shape)
    for d in dec.domains:
        r = sp.csr_matrix((np.ones(len(d.l2g)), (np.arange(len(d.l2g)), d.l2g)), shape=(len(d.l2g), a.shape[0]))
        acc = acc + r.T @ d.a_neu @ r
    assert abs(acc - a).max() < 1e-12


def test_partitioner_handles_disconnected_and_tiny_graphs():
    g = sp.block_diag([sp.csr_matrix(np.ones((3, 3)) - np.eye(3)), sp.csr_matrix(np.ones((4, 4)) - np.eye(4))]).tocsr()
    p = decomp.partition_graph(g, 3)
    assert sorted(np.bincount(p).tolist()) == [2, 2, 3]
    with pytest.raises(ValueError):
        decomp.partition_graph(g, 8)


@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("n,k", [(10, 8), (9, 5), (20, 8)])
def test_native_kway_partitioner(n, k, dual):
    """The library's C++ partitioner (csrc/partition.cpp, GeneoPartMeshDual / Nodal behind the C ABI: the counterparts of
    the driver's METIS_PartMeshDual / METIS_PartMeshNodal calls, src/geneo4PETSc.cpp:381-445): sizes exact to one vertex
    per bisection, every part non-empty, deterministic, the cut of the structured 2x2x2 blocks on a cube (the optimum)
    and within 1.1x of the numpy / scipy prototype it was ported from; the second output follows the first."""
    mesh = decomp.grid_mesh(size=n, dim=3)
    g = decomp.mesh_graph(mesh, dual)
    ep, npt, cut = decomp.partition_mesh_native(mesh, k, dual)
    p = ep if dual else npt
    sizes = np.bincount(p, minlength=k)
    assert sizes.min() >= 1 and sizes.max() - sizes.min() <= max(2, int(np.ceil(np.log2(k))) + 1)
    assert cut == decomp.edge_cut(g, p)
    ep2, npt2, cut2 = decomp.partition_mesh_native(mesh, k, dual)
    assert np.array_equal(ep, ep2) and np.array_equal(npt, npt2) and cut == cut2
    assert cut <= 1.1 * decomp.edge_cut(g, decomp.partition_graph(g, k))
    if not dual and k == 8:
        nn = decomp.grid_size(n, 1, 3)
        assert cut <= 1.05 * decomp.edge_cut(g, decomp.structured_node_partition(nn, 3, (2, 2, 2)))
    # the other kind of object follows the partitioned one (a node the part of one of its elements, and vice versa)
    e, w = np.nonzero(mesh.nodes >= 0)
    if dual:
        ok = np.zeros(mesh.nbNode, dtype=bool)
        ok[mesh.nodes[e, w][ep[e] == npt[mesh.nodes[e, w]]]] = True
        assert ok.all()
    else:
        assert (ep == npt[mesh.nodes[:, 0]]).all()
    dec = decomp.decompose(mesh, k, ep if dual else None, None if dual else npt, dual, 1)
    assert len(dec.domains) == k and all(len(d.l2g) > 0 for d in dec.domains)


def test_native_partitioner_irregular_graph_and_csr_entry():
    """tst/graph input (configs[4] in small): GeneoPartGraphKway on the nodal graph == GeneoPartMeshNodal on the mesh;
    one part and more parts than vertices behave as the Metis wrapper of the driver expects (driver:397-400)."""
    mesh = decomp.graph_mesh(size=900, level=2, no_ground=True)
    g = decomp.mesh_graph(mesh, False)
    _, npt, cut = decomp.partition_mesh_native(mesh, 8, False)
    p = decomp.partition_graph_native(g, 8)
    assert np.array_equal(p, npt) and decomp.edge_cut(g, p) == cut
    sizes = np.bincount(p, minlength=8)
    assert sizes.max() - sizes.min() <= 4
    assert cut <= 1.1 * decomp.edge_cut(g, decomp.partition_graph(g, 8))
    assert (decomp.partition_graph_native(g, 1) == 0).all()
    with pytest.raises(RuntimeError):
        decomp.partition_graph_native(g, g.shape[0] + 1)


def test_native_generator_and_decomposition_equal_the_prototypes():
    """csrc/decompose.cpp (GeneoGridMesh, GeneoDecompCreate / GeneoDecompDomain: the C++ counterparts of the reference's
    generators and of driver:196-379, :447-494, :643-715, used by bench.py) against the numpy prototypes the parity tests
    read: generator output identical to the bit (elements, order, values: laplacian / heat, every interpolation, 1-D to
    3-D, windowed); decomposition lists identical, matrix patterns identical, values to a rounding error of the summation
    order; the windowed one-domain path too."""
    for kw in (dict(n=9, dim=3), dict(n=12, dim=2, kappa_max=3.0, interp="lin"), dict(n=8, dim=1),
               dict(n=7, dim=3, heat=True, kappa_max=100.0, interp="minmax"),
               dict(n=9, dim=3, kappa_max=2.0, interp="quad", window=((1, 0, 2), (7, 9, 8)))):
        a, b = decomp.grid_mesh(**kw), decomp.grid_mesh_native(**kw)
        assert a.nbNode == b.nbNode and np.array_equal(a.nodes, b.nodes) and np.array_equal(a.mats, b.mats), kw
    mesh = decomp.grid_mesh(n=10, dim=3, kappa_max=2.0, interp="lin")
    for dual in (False, True):
        ep, npt = decomp.partition_mesh(mesh, 5, dual)
        for ov in (0, 2):
            ref = decomp.decompose(mesh, 5, ep, npt, dual, ov).domains
            nat = decomp.decompose_native(mesh, 5, ep, npt, dual, ov)
            for r, t in zip(ref, nat):
                assert np.array_equal(r.l2g, t.l2g) and np.array_equal(r.mult, t.mult)
                for x, y in ((r.a_neu, t.a_neu), (r.a_dir, t.a_dir)):
                    assert np.array_equal(x.indptr, y.indptr) and np.array_equal(x.indices, y.indices)
                    assert np.abs(x.data - y.data).max() <= 4e-16 * np.abs(x.data).max()
                assert all(np.array_equal(i, j) for i, j in zip(r.intersect, t.intersect))
    d1 = decomp.decompose_grid_domain(20, 3, (2, 2, 2), 2, 5)
    d2 = decomp.decompose_grid_domain(20, 3, (2, 2, 2), 2, 5, native=True)
    assert np.array_equal(d1.l2g, d2.l2g) and np.array_equal(d1.mult, d2.mult) and abs(d1.a_neu - d2.a_neu).max() < 1e-15
    assert abs(d1.a_dir - d2.a_dir).max() < 1e-15 and all(np.array_equal(i, j) for i, j in zip(d1.intersect, d2.intersect))
