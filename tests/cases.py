"""Shared test-case builders (product host code for the inputs, oracle for the expected values)."""
import os

import numpy as np

from geneo4petsc_amd import decomp
from oracle import geneo_oracle as go


def grid_case(n=12, dim=3, parts=(2, 2, 2), overlap=1, **gen):
    mesh = decomp.grid_mesh(n=n, dim=dim, **gen)
    nb = parts[0] * parts[1] * parts[2]
    dec = decomp.decompose(mesh, nb, None, decomp.structured_node_partition(n, dim, parts), False, overlap)
    a = decomp.global_matrix(mesh)
    b = decomp.rhs_default(a)
    return mesh, dec, a, b


def oracle_for(mesh, dec, argv, b, literal=False, arpack_seed=None):
    """literal=False: eigenpairs exact (the limit -els2_eps_tol -> 0).  literal=True: the reference's own call, ARPACK
    shift-invert AT -els2_eps_tol (geneo.cpp:649-663) -- the oracle of the parity tests at the bench option set, where
    the tolerance is the reference's default 1e-3 and the quality of the deflation depends on it."""
    subs = [go.Subdomain(d.l2g, d.a_neu, d.mult, d.intersect) for d in dec.domains]
    orc = go.GenEOOracle(mesh.nbNode, subs, go.parse_options(argv))
    if literal:
        orc.dense_limit, orc.exact_eigs = 0, False
        if arpack_seed is not None:
            orc.arpack_seed = arpack_seed
    else:
        orc.dense_limit = 4000      # LAPACK ground truth for every test-sized pencil (ARPACK misses multiplets)
        orc.exact_eigs = True       # above that: ARPACK to machine precision + inertia proof that no copy is missing
    return orc.setup(b)


class Tight:
    """`argv + TIGHT`: the tolerances of the mode-parity tests, chosen by the Krylov method already in argv.
    Eigenpairs converged to 1e-10 (the limit in which the coarse space is unique).  Krylov tolerance 1e-8 for GMRES;
    1e-6 for CG: PCG amplifies the rounding-level difference between ANY two implementations of the same
    preconditioner (exact LU here, PCG to 1e-12 there; or two different summation orders) about tenfold per iteration
    once its first Ritz values have converged -- 1e-11 at iteration 12, 1e-6 at 17, O(1) near 25 on the 12^3 cases
    (tests/test_oracle_eig.py::test_cg_count_is_rounding_sensitive shows it on the oracle alone).  Counts are compared
    where the two runs are still the same computation, and there they must be IDENTICAL: no tolerance on the count."""

    def __radd__(self, argv):
        ksp = "gmres"
        for i, a in enumerate(argv):
            if a == "-ksp_type":
                ksp = argv[i + 1]
        return list(argv) + ["-els2_eps_tol", "1e-10", "-ksp_rtol", "1e-6" if ksp == "cg" else "1e-8"]


TIGHT = Tight()
BENCH_OVERLAP = 2


def bench_argv(extra=()):
    """bench.py's own option set (bench.py::geneo_argv with its defaults; tests/test_bench_options.py keeps the two
    in step)."""
    return ["-geneo_lvl", "SRAS,1", "-geneo_tau", "0.35", "-geneo_cut", "20", "-els2_eps_tol", "0.001",
            "-ksp_type", "cg", "-ksp_rtol", "1e-05", "-dls1_ksp_rtol", "1e-07", "-dls1_pc_type", "amg",
            "-els2_pc_type", "amg"] + list(extra)


def ksp_args(argv):
    out = dict(rtol=1e-5, atol=1e-50, max_it=10000, restart=30)
    ksp = "gmres"
    for i, a in enumerate(argv):
        if a == "-ksp_type":
            ksp = argv[i + 1]
        if a == "-ksp_rtol":
            out["rtol"] = float(argv[i + 1])
        if a == "-ksp_atol":
            out["atol"] = float(argv[i + 1])
        if a == "-ksp_gmres_restart":
            out["restart"] = int(argv[i + 1])
    if ksp == "cg":
        out.pop("restart")
    return ksp, out


def run_pc(lib, mesh, dec, argv, b, with_dir=True, with_intersect=False):
    from geneo4petsc_amd.pc import GenEOPC
    pc = GenEOPC(lib)
    pc.set_from_options(argv)
    pc.set_sizes(mesh.nbNode, len(dec.domains))
    for d in dec.domains:
        pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir if with_dir else None)
        if with_intersect:
            pc.set_intersect(d.gid, [len(x) > 0 for x in d.intersect])
    pc.setup(b)
    return pc


def graph_case(size=400, level=2, nb=4, overlap=1, no_ground=True):
    """BASELINE config 5 in small: tst/graph generator, node-range partition, irregular CSR."""
    mesh = decomp.graph_mesh(size=size, level=level, no_ground=no_ground)
    dec = decomp.decompose(mesh, nb, None, decomp.graph_node_partition(mesh, nb), False, overlap)
    a = decomp.global_matrix(mesh)
    return mesh, dec, a, decomp.rhs_default(a)


RECORD_MODE_HITS = []  # mismatches that GENEO_TEST_COUNT_DRIFT=record let through: a discovery run never ends green
COUNT_DRIFT = []     # (case, library count, oracle count) of every declared / recorded iteration-count difference


def compare_with_oracle(lib, n, parts, overlap, argv, gen=None, with_dir=True, xtol=1e-8, aptol=1e-9, case=None,
                        dim=3, with_intersect=False, count_drift=None):
    """Full parity check of one configuration: integer outputs exact, floats within tolerance."""
    if case is not None:
        mesh, dec, a, b = case
    else:
        mesh, dec, a, b = grid_case(n=n, dim=dim, parts=parts, overlap=overlap, **(gen or {}))
    pc = run_pc(lib, mesh, dec, argv, b, with_dir, with_intersect)
    orc = oracle_for(mesh, dec, argv, b)
    ksp, kw = ksp_args(argv)
    res = go.solve(orc, b, ksp, **kw)
    x, its, rnorm, reason = pc.solve(b)
    info = pc.info()
    assert pc.name == orc.o.name
    assert reason == res.reason, (reason, res.reason)
    # Iteration count: identical to the oracle's -- always for GMRES, and for CG unless the case is DECLARED
    # rounding-sensitive by its test (count_drift=(library count, oracle count), justified there).  Every use of
    # such a declaration is recorded in COUNT_DRIFT and printed in the terminal summary (tests/conftest.py), so a
    # run shows which cases took it; an undeclared difference fails.  GENEO_TEST_COUNT_DRIFT=record lets the run
    # continue past such a failure to collect all of them (discovery on a new device); the SESSION still ends red
    # (tests/conftest.py::pytest_sessionfinish).
    if its != res.its:
        label = "%s n=%s parts=%s overlap=%s %s" % (lib.GeneoBackendName().decode(), n, parts, overlap, " ".join(argv))
        COUNT_DRIFT.append((label, its, res.its))
        k = min(8, len(res.history), len(pc.residual_history()))
        np.testing.assert_allclose(pc.residual_history()[:k], res.history[:k], rtol=1e-8)
        if os.environ.get("GENEO_TEST_COUNT_DRIFT") == "record":
            RECORD_MODE_HITS.append(label)       # tests/conftest.py turns the session red when this list is not empty
        else:
            assert ksp == "cg", "GMRES iteration count differs: %d vs oracle %d" % (its, res.its)
            assert count_drift is not None, "undeclared CG iteration-count difference: %d vs oracle %d" % (its, res.its)
            assert (its, res.its) in count_drift, "CG count %d vs oracle %d is not the declared drift %r" % (
                its, res.its, count_drift)
    if orc.o.lvl2:
        assert list(pc.local_dims()) == orc.realDimELoc                                  # integer selection
        assert info["nicolaidesLoc"] == sum(orc.nicolaidesLoc)
        assert info["dimE"] == orc.dimE
        if orc.o.lvl2 == 2 and not orc.o.cst:                                            # geneo.cpp:1097-1232
            t, g = pc.local_params()
            np.testing.assert_allclose(t, orc.tauLoc, rtol=1e-14)
            np.testing.assert_allclose(g, orc.gammaLoc, rtol=1e-12)
        for s in range(len(dec.domains)):
            ev = np.sort(pc.eigenvalues(s))
            np.testing.assert_allclose(ev, np.sort(orc.eigvals[s]), rtol=1e-10, atol=1e-13)  # 1e-10 relative
        q1, q2 = pc.apply_q(b), orc.apply_q(b)
        assert np.linalg.norm(q1 - q2) <= aptol * np.linalg.norm(q2)
    y1, y2 = pc.apply(b), orc.apply(b)
    assert np.linalg.norm(y1 - y2) <= aptol * np.linalg.norm(y2)
    m1, m2 = pc.matmult(b), orc.matmult(b)
    assert np.linalg.norm(m1 - m2) <= 1e-13 * np.linalg.norm(m2)
    # same iterate when the counts agree; otherwise two different iterates of the same convergent sequence,
    # each within the Krylov tolerance of the solution
    # (PCG: the two iterates carry the amplified rounding difference described in Tight -- up to 1e-7 after ~20
    # iterations on the GPU, whose summation orders differ from numpy's: bar = the Krylov tolerance 1e-6 itself)
    rel = np.linalg.norm(x - res.x) / np.linalg.norm(res.x)
    bar = (xtol if its == res.its else 5 * xtol) * (100.0 if ksp == "cg" else 1.0)
    assert rel <= bar, "iterate differs from the oracle's by %.2e (bar %.1e)" % (rel, bar)
    pc.destroy()
    return its, info


def check_singular_neumann_case(lib, n, argv):
    """--inpEps 0: the Neumann matrices of the four subdomains away from the Dirichlet face are exactly singular.  What
    the reference does there depends on the sign of a rounding error: its Nicolaides rule (geneo.cpp:897-944) looks at
    min(lambda) >= DBL_EPSILON, and a zero eigenvalue computed as +2e-15 (LAPACK does that on two of the four floating
    subdomains at 20^3) makes it add the constant vector a second time -- a rank-deficient Z.  So the oracle's COUNT is
    not a ruler here; compared are: the non-zero eigenvalues, exactly one (numerically) zero eigenvalue per floating
    subdomain, null pivots detected (tuneSolver, geneo.cpp:76-92), and a converged solve with the right solution."""
    mesh, dec, a, b = grid_case(n=n, dim=3, parts=(2, 2, 2), overlap=1, inp_eps=0.0)
    pc = run_pc(lib, mesh, dec, argv, b)
    orc = oracle_for(mesh, dec, argv, b)
    info = pc.info()
    assert info["nullPivotsLoc"] >= 4, info["nullPivotsLoc"]
    floating = 0
    for s in range(len(dec.domains)):
        ev, ov = np.sort(pc.eigenvalues(s)), np.sort(orc.eigvals[s])
        zl, zo = ev[np.abs(ev) < 1e-10], ov[np.abs(ov) < 1e-10]
        assert len(zl) == (1 if len(zo) else 0), (s, ev[:3], ov[:3])      # the oracle may hold its zero twice (see above)
        floating += len(zl)
        nl, no = ev[np.abs(ev) >= 1e-10], ov[np.abs(ov) >= 1e-10]
        k = min(len(nl), len(no))
        assert k >= 1
        np.testing.assert_allclose(nl[:k], no[:k], rtol=1e-9)
    assert floating == 4
    x, its, rnorm, reason = pc.solve(b)
    assert reason == "KSP_CONVERGED_RTOL" and its < 40
    xs = np.arange(1.0, mesh.nbNode + 1.0)
    assert np.linalg.norm(x - xs) <= 1e-6 * np.linalg.norm(xs)
    pc.destroy()
    return info


def check_grouped_eigensolve(lib, n=14, overlap=2, extra=(), group_rows=(1, 3000), exact=True):
    """Memory-bounded set-up (-geneo_eig_group_rows): the rank's subdomains eigensolved in consecutive groups, each with its
    own A_Neu hierarchy and LOBPCG blocks.  A subdomain's iteration depends on nothing outside its own rows (start block
    from its global id and local row, its own Gershgorin bounds in the hierarchy, per-subdomain Rayleigh-Ritz and
    freezing), so every INTEGER the set-up produces -- kept counts, dimE, Nicolaides, the LOBPCG iteration count -- and the
    PCG count of the solve that follows are those of the all-at-once path.
    exact=True (host simulator: one kernel form per operation): eigenvalues, E, the residual history and the solution are
    BIT-IDENTICAL.  exact=False (the GPU: the SpMV form of a coarse level -- slices or lanes-per-row, two summation
    orders -- is picked from the row count of the whole batch): floats agree to rounding, eigenvalues to 1e-10 relative
    (north_star's bar), the worst differences are returned for the test to print."""
    mesh, dec, a, b = grid_case(n=n, dim=3, parts=(2, 2, 2), overlap=overlap)
    argv = bench_argv(extra)
    runs = []
    for rows in (0,) + tuple(group_rows):
        pc = run_pc(lib, mesh, dec, argv + (["-geneo_eig_group_rows", str(rows)] if rows else []), b)
        x, its, rnorm, reason = pc.solve(b)
        info = pc.info()
        runs.append(dict(ev=[pc.eigenvalues(s) for s in range(8)], cand=[pc.eigenvalues(s, candidates=True) for s in range(8)],
                         E=pc.E(), dims=list(pc.local_dims()), its=its, x=x, reason=reason, info=info, hist=pc.residual_history()))
        pc.destroy()
    base = runs[0]
    assert base["info"]["eigGroups"] == 1 and base["reason"].startswith("KSP_CONVERGED")
    worst = {"eigenvalue": 0.0, "E": 0.0, "x": 0.0}
    for rows, r in zip(group_rows, runs[1:]):
        assert r["info"]["eigGroups"] > 1, (rows, r["info"]["eigGroups"])
        assert r["dims"] == base["dims"] and r["info"]["dimE"] == base["info"]["dimE"]
        assert r["info"]["eig_iterations"] == base["info"]["eig_iterations"]
        assert r["info"]["nicolaidesLoc"] == base["info"]["nicolaidesLoc"]
        assert r["its"] == base["its"] and r["reason"] == base["reason"], (rows, r["its"], base["its"])
        for s in range(8):
            if exact:
                assert np.array_equal(r["ev"][s], base["ev"][s]), (rows, s)
                assert np.array_equal(r["cand"][s], base["cand"][s]), (rows, s)
            else:
                np.testing.assert_allclose(r["ev"][s], base["ev"][s], rtol=1e-10, atol=1e-13)
                worst["eigenvalue"] = max(worst["eigenvalue"], float(np.max(np.abs(r["ev"][s] - base["ev"][s]) / np.abs(base["ev"][s]))))
        if exact:
            assert np.array_equal(r["E"], base["E"]), rows
            assert np.array_equal(r["hist"], base["hist"]) and np.array_equal(r["x"], base["x"]), rows
        else:
            worst["E"] = max(worst["E"], float(np.linalg.norm(r["E"] - base["E"]) / np.linalg.norm(base["E"])))
            worst["x"] = max(worst["x"], float(np.linalg.norm(r["x"] - base["x"]) / np.linalg.norm(base["x"])))
            assert worst["E"] <= 1e-8 and worst["x"] <= 1e-6, worst
    assert runs[1]["info"]["eigGroups"] == 8
    return runs, worst


def check_coarse_start(lib, n, extra=(), ev_rtol=1e-8, expect_levels=()):
    """-geneo_eig_coarse_start (the local eigensolve started from the Ritz vectors of the multigrid level-1 pencil, that one
    from level 2, ...): a start block, nothing else -- at a tight eigenvalue tolerance the kept eigenvalues, the kept counts,
    dimE and the outer iteration count are those of the seeded random start.  Returns (info without, info with, worst
    relative eigenvalue difference)."""
    mesh, dec, a, b = grid_case(n=n, dim=3, parts=(2, 2, 2), overlap=2)
    runs = []
    for cs in (0, 1):
        pc = run_pc(lib, mesh, dec, bench_argv(["-els2_eps_tol", "1e-10", "-geneo_eig_coarse_start", str(cs)] + list(extra)), b)
        x, its, rnorm, reason = pc.solve(b)
        runs.append(dict(ev=[pc.eigenvalues(s) for s in range(8)], dims=list(pc.local_dims()), its=its, x=x, reason=reason,
                         info=pc.info()))
        pc.destroy()
    off, on = runs
    assert off["info"]["eigCoarseIterations"] == 0 and on["info"]["eigCoarseIterations"] > 0
    assert on["reason"].startswith("KSP_CONVERGED") and on["its"] == off["its"], (on["its"], off["its"])
    assert on["dims"] == off["dims"] and on["info"]["dimE"] == off["info"]["dimE"]
    worst = 0.0
    for s in range(8):
        np.testing.assert_allclose(on["ev"][s], off["ev"][s], rtol=ev_rtol, atol=1e-13)
        worst = max(worst, float(np.max(np.abs(on["ev"][s] - off["ev"][s]) / np.abs(off["ev"][s]))))
    assert np.linalg.norm(on["x"] - off["x"]) <= 1e-6 * np.linalg.norm(off["x"])
    return off["info"], on["info"], worst
