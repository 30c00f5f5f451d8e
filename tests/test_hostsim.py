"""Host logic of libgeneopc (core.cpp / capi.cpp: options, layout, LOBPCG driver, batched-CG driver,
E assembly, apply modes, Krylov loops) exercised on CPU through the TEST-ONLY serial backend
(tests/hostsim) and compared with the oracle.  The HIP kernels themselves are tested in -m gpu."""
import numpy as np
import pytest

import cases
import dummy_cases as dc
import hostsim_util as hu

TIGHT = cases.TIGHT      # eigenpairs 1e-10; Krylov 1e-8 (GMRES) / 1e-6 (CG): see cases.Tight


@pytest.fixture(scope="module")
def lib():
    return hu.hostsim_lib()


@pytest.mark.parametrize("lvl,ksp,overlap", [("ASM,0", "cg", 1), ("ASM,1", "cg", 1), ("RAS,1", "gmres", 1),
                                             ("SRAS,H1", "cg", 2), ("ASM,E1", "gmres", 1), ("ORAS,1", "gmres", 1)])
def test_modes(lib, lvl, ksp, overlap):
    argv = ["-geneo_lvl", lvl, "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", ksp, "-geneo_optim", "0.02"] + TIGHT
    cases.compare_with_oracle(lib, 12, (2, 2, 2), overlap, argv)


@pytest.mark.parametrize("lvl,ksp,n,parts,cut,inter", [("SORAS,2", "cg", 12, (3, 2, 1), 12, True),
                                                       ("ORAS,H2", "gmres", 10, (2, 2, 2), 10, False),
                                                       ("SORAS,E2", "cg", 6, (2, 2, 1), 12, False)])
def test_geneo2(lib, lvl, ksp, n, parts, cut, inter):
    """GenEO-2 (geneo.cpp:1274-1300): tau problem on (A_Neu, A_Rob) with tau_loc, gamma problem on (D A_Dir D, A_Rob)
    with gamma_loc from the connectivity matrix, -geneo_cut halved.  gamma 1.05 makes the gamma problem contribute
    vectors; n = 6 runs the dense small-subdomain path, the others LOBPCG (inverted pencil for the largest ones)."""
    argv = ["-geneo_lvl", lvl, "-geneo_tau", "0.02", "-geneo_gamma", "1.05", "-geneo_cut", str(cut), "-geneo_optim", "0.5",
            "-ksp_type", ksp] + TIGHT          # cut chosen so that cut/2 does not fall inside a multiplet
    _, info = cases.compare_with_oracle(lib, n, parts, 1, argv, with_intersect=inter)
    assert info["dimE"] > len(parts)


def test_geneo2_cst_and_chebyshev_fallback(lib):
    """Jacobi-PCG local solves reach an operator accuracy of 5e-12 (the oracle's LU: 1e-16).  On this SORAS case PCG
    amplifies that difference tenfold per iteration from iteration 13 on (1e-11 at 12, 1e-6 at 17, O(1) at 21): the
    count is therefore taken at -ksp_rtol 1e-6 (iteration 17, histories still equal to 1e-6), not at 1e-8 where the
    library needed 23 iterations against the oracle's 21 in round 1 -- identical counts, no tolerance on them."""
    argv = ["-geneo_lvl", "SORAS,2", "-geneo_tau", "0.05", "-geneo_gamma", "1.2", "-geneo_cst", "-geneo_cut", "8",
            "-geneo_optim", "0.1", "-ksp_type", "cg", "-els2_pc_type", "cheb", "-dls1_pc_type", "jacobi",
            "-els2_eps_tol", "1e-10", "-ksp_rtol", "1e-6"]
    cases.compare_with_oracle(lib, 10, (2, 2, 1), 1, argv, xtol=1e-6)


def test_multilevel_amg_inner_preconditioner(lib):
    """Two-level smoothed-aggregation hierarchy inside the local PCG and inside LOBPCG: same outer
    operator as the oracle's exact LU (parity unchanged), far fewer inner iterations than Jacobi."""
    base = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.19", "-geneo_cut", "8", "-ksp_type", "cg"] + TIGHT
    _, amg = cases.compare_with_oracle(lib, 20, (2, 2, 2), 1, base + ["-amg_coarse_size", "200"])
    _, jac = cases.compare_with_oracle(lib, 20, (2, 2, 2), 1, base + ["-dls1_pc_type", "jacobi", "-els2_pc_type", "cheb"])
    assert amg["amg_levels"] >= 2 and jac["amg_levels"] == 0
    assert amg["dls1_iterations"] * 3 < jac["dls1_iterations"]


def test_threaded_aggregation_gives_the_serial_aggregates(lib, monkeypatch):
    """Aggregation with its strong-neighbour lists and its leftover pass in row ranges on several host threads (what one
    big subdomain per GPU gets) == the one-thread walk: same hierarchy sizes, same inner and outer iteration counts."""
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.19", "-geneo_cut", "8", "-ksp_type", "cg", "-amg_coarse_size", "100"] + TIGHT
    _, one = cases.compare_with_oracle(lib, 20, (2, 2, 2), 1, argv)
    monkeypatch.setenv("GENEO_AGG_THREADS", "3")
    monkeypatch.setenv("GENEO_AGG_MIN_NNZ", "0")
    _, three = cases.compare_with_oracle(lib, 20, (2, 2, 2), 1, argv)
    for key in ("amg_levels", "amg_operator_complexity", "dls1_iterations", "eig_iterations", "dimE"):
        assert one[key] == three[key], key


def test_config0_laplacian_2d_two_subdomains_five_vectors(lib):
    """BASELINE configs[0]: tst/laplacian 2-D stencil, 2 subdomains, 5 eigenvectors per subdomain."""
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.9", "-geneo_cut", "5", "-ksp_type", "gmres"] + TIGHT
    its, info = cases.compare_with_oracle(lib, 40, (2, 1, 1), 1, argv, dim=2, gen=dict(kappa_max=2.0, interp="lin"))
    assert info["dimE"] == 10


@pytest.mark.parametrize("no_ground", [True, False])
def test_config4_graph_irregular_csr(lib, no_ground):
    """BASELINE configs[4] in small: tst/graph generator (with the ground node the matrix has one very
    long row -> long-row SpMV kernel), node-range partition, 4 subdomains."""
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.3", "-geneo_cut", "6", "-ksp_type", "cg"] + TIGHT
    cases.compare_with_oracle(lib, 0, None, 1, argv, case=cases.graph_case(size=400, level=2, nb=4, overlap=1,
                                                                           no_ground=no_ground))


def test_dirichlet_built_from_matis(lib):
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", "cg"] + TIGHT
    cases.compare_with_oracle(lib, 10, (2, 2, 1), 1, argv, with_dir=False)


def test_option_errors(lib):
    from geneo4petsc_amd.pc import GenEOPC, GenEOError
    pc = GenEOPC(lib)
    assert pc.name == "geneo1ASM"                         # defaults, geneo.cpp:2649-2651
    with pytest.raises(GenEOError, match="invalid option -geneo_lvl, unknown FOO"):
        pc.set_from_options(["-geneo_lvl", "FOO,1"])
    with pytest.raises(GenEOError, match="tau must be < 1"):
        pc.set_from_options(["-geneo_tau", "1.5"])
    pc2 = GenEOPC(lib)
    pc2.set_from_options(["-geneo_lvl", "SORAS,H1", "-pc_type", "geneo", "-unknown_flag"])
    assert pc2.name == "geneo1HSORAS"
    pc3 = GenEOPC(lib)
    pc3.set_from_options(["-geneo_lvl", "ASM,2"])
    with pytest.raises(GenEOError):
        pc3.set_sizes(4, 1)
        pc3.setup()                                       # no subdomain: loud error
    mesh, dec, a, b = cases.grid_case(6, 3, (2, 1, 1), 1)
    with pytest.raises(GenEOError, match="GenEO-2 needs the Robin matrix"):
        cases.run_pc(lib, mesh, dec, ["-geneo_lvl", "ASM,2"], b)


@pytest.mark.parametrize("rec", dc.geneo_refs()[::3],
                         ids=lambda r: r["file"][:-4])
def test_dummy_goldens(lib, rec):
    from geneo4petsc_amd import decomp
    from geneo4petsc_amd.pc import GenEOPC
    d = dc.load()
    mesh = decomp.read_input_text(d["inputs"][rec["input"] + ".inp"], rec["inpEps"])
    ep, npart = dc.partition_for(rec)
    dec = decomp.decompose(mesh, 2, ep, npart, rec["metis"] == "dual", rec["overlap"])
    a = decomp.global_matrix(mesh)
    b = decomp.read_b_text(d["inputs"]["B.inp"], mesh.nbNode) if rec["use_b_file"] else decomp.rhs_default(a)
    argv = ["-geneo_lvl", rec["geneo_lvl"], "-ksp_rtol", "1e-12", "-ksp_atol", "1e-12"]
    if rec["geneo_cut"] > 0:
        argv += ["-geneo_cut", str(rec["geneo_cut"])]
    pc = GenEOPC(lib)
    pc.set_from_options(argv)
    assert ("INFO: %s pc" % pc.name) in rec["info"][2]
    pc.set_sizes(mesh.nbNode, 2)
    for dom in dec.domains:
        pc.add_subdomain(dom.gid, dom.l2g, dom.mult, dom.a_neu, None)
    pc.setup(b)
    x, its, rnorm, reason = pc.solve(b)
    assert reason.startswith("KSP_CONVERGED")
    np.testing.assert_allclose(x, rec["x"], rtol=1e-5, atol=1e-6)


def test_known_answer_E_and_Qb(lib):
    """tridiag / dual: E = diag(10, 8), Qb = (2,2,2,4.25,6.5,6.5,6.5,6.5), nicolaides 2."""
    from geneo4petsc_amd import decomp
    from geneo4petsc_amd.pc import GenEOPC
    rec = [r for r in dc.geneo_refs() if r["file"] == "tridiag-pc=geneoASM1-metis=dual.ref"][0]
    d = dc.load()
    mesh = decomp.read_input_text(d["inputs"]["tridiag.inp"], 1.0)
    ep, npart = dc.partition_for(rec)
    dec = decomp.decompose(mesh, 2, ep, npart, True, 0)
    b = decomp.rhs_default(decomp.global_matrix(mesh))
    pc = GenEOPC(lib)
    pc.set_from_options(["-geneo_lvl", "ASM,1"])
    pc.set_sizes(8, 2)
    for dom in dec.domains:
        pc.add_subdomain(dom.gid, dom.l2g, dom.mult, dom.a_neu, None)
    pc.setup(b)
    info = pc.info()
    assert info["nicolaidesLoc"] == 2 and info["dimE"] == 2 and list(pc.local_dims()) == [1, 1]
    np.testing.assert_allclose(pc.E(), np.diag([10.0, 8.0]), atol=1e-12)
    np.testing.assert_allclose(pc.apply_q(b), [2, 2, 2, 4.25, 6.5, 6.5, 6.5, 6.5], atol=1e-12)


def test_geneo_chk_diagnostics(lib, tmp_path, monkeypatch):
    """-geneo_chk (geneo.cpp:173-247, :782-840, :988-997): SPD logs of A and of each pencil's B, R of Z = QR
    locally and globally, under the reference's file names; results unchanged by the checks."""
    monkeypatch.chdir(tmp_path)
    argv = ["-geneo_lvl", "SORAS,2", "-geneo_tau", "0.02", "-geneo_gamma", "1.05", "-geneo_cut", "10", "-geneo_optim", "0.5",
            "-ksp_type", "cg", "-geneo_chk", "log"] + TIGHT
    _, info = cases.compare_with_oracle(lib, 10, (2, 2, 2), 1, argv)
    a_log = (tmp_path / "check.SPD.A.log").read_text()
    lmin = float(a_log.splitlines()[0].split(":")[1])
    mesh, dec, a, b = cases.grid_case(10, 3, (2, 2, 2), 1)
    import scipy.sparse.linalg as spla
    true_min = spla.eigsh(a, k=1, sigma=0, which="LM")[0][0]
    assert 0 < true_min <= lmin * (1 + 1e-12) and lmin < 1.2 * true_min       # Lanczos Ritz value: upper bound, close
    for gid in range(8):
        for pb in ("tau", "gamma"):
            txt = (tmp_path / ("check%d.SPD.%s.B.log" % (gid, pb))).read_text()
            assert txt.startswith(pb + ".B - eigen value 0: ") and "nbNegEV 0, nbNullEV 0, nbPosEV 290" in txt
        r = np.loadtxt(tmp_path / ("check%d.setup.Z.R" % gid))
        assert r.shape[0] == r.shape[1] and np.all(np.abs(np.diag(r)) > 1e-8) and np.allclose(r, np.triu(r))
    rg = np.loadtxt(tmp_path / "check.setup.ZE2G.R")
    assert rg.shape == (info["dimE"], info["dimE"]) and np.all(np.diag(rg) > 0)
    # small subdomains (dense path) write the same files
    cases.compare_with_oracle(lib, 6, (2, 1, 1), 1, ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-geneo_cut", "4", "-geneo_chk", "log"] + TIGHT)
    assert "nbNegEV 0, nbNullEV 0" in (tmp_path / "check1.SPD.tau.B.log").read_text()


def test_geneo_chk_flags_a_singular_operator(lib, tmp_path, monkeypatch):
    from geneo4petsc_amd import decomp
    from geneo4petsc_amd.pc import GenEOPC, GenEOError
    monkeypatch.chdir(tmp_path)
    d = dc.load()
    mesh = decomp.read_input_text(d["inputs"]["tridiag.inp"], 1.0)
    width = mesh.nodes.shape[1]
    for k in range(len(mesh.mats)):                         # pure Neumann chain: singular A
        if np.count_nonzero(mesh.nodes[k] >= 0) == 2:
            m = np.zeros((width, width))
            m[:2, :2] = [[1.0, -1.0], [-1.0, 1.0]]
            mesh.mats[k] = m.ravel()
        else:
            mesh.mats[k] = 0.0
    rec = [r for r in dc.geneo_refs() if r["file"] == "tridiag-pc=geneoASM1-metis=dual.ref"][0]
    ep, npart = dc.partition_for(rec)
    dec = decomp.decompose(mesh, 2, ep, npart, True, 0)
    pc = GenEOPC(lib)
    pc.set_from_options(["-geneo_lvl", "ASM,1", "-geneo_chk", "log"])
    with pytest.raises(GenEOError, match="invalid option -geneo_chk, unknown foo"):
        GenEOPC(lib).set_from_options(["-geneo_chk", "foo"])
    pc.set_sizes(8, 2)
    for dom in dec.domains:
        pc.add_subdomain(dom.gid, dom.l2g, dom.mult, dom.a_neu, None)
    with pytest.raises(GenEOError, match="GenEO - check SPD: A not SPD"):
        pc.setup(np.ones(8))


def test_no_cut_keeps_every_eigenvalue_below_tau(lib):
    """Without -geneo_cut the reference's nev is the inertia count (geneo.cpp:502-560): every eigenvalue below tau
    enters Z.  The LOBPCG block grows (16 -> 32 -> 64 columns) until the threshold falls inside it; small
    subdomains read the count off the dense spectrum."""
    _, info = cases.compare_with_oracle(lib, 10, (2, 2, 2), 1, ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.42", "-ksp_type", "cg"] + TIGHT)
    assert info["dimE"] == 276
    cases.compare_with_oracle(lib, 6, (2, 1, 1), 1, ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.3", "-ksp_type", "cg"] + TIGHT)


def test_E_blocked_and_columnwise_assembly_agree(lib, monkeypatch):
    """E = Z^T A Z: 32 columns per pass (MFMA block kernels, wide halo exchanges) vs one column per pass (hosts
    whose transport holds one vector per exchange, PCGenEOSetCommWidth not called)."""
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-geneo_cut", "8"] + TIGHT
    mesh, dec, a, b = cases.grid_case(10, 3, (2, 2, 2), 1)
    pc = cases.run_pc(lib, mesh, dec, argv, b)
    e_blocked = pc.E().copy()
    pc.destroy()
    monkeypatch.setenv("GENEO_E_COLUMNWISE", "1")
    pc = cases.run_pc(lib, mesh, dec, argv, b)
    e_col = pc.E().copy()
    pc.destroy()
    orc = cases.oracle_for(mesh, dec, argv, b)
    assert e_blocked.shape == e_col.shape == orc.E.shape
    np.testing.assert_allclose(e_blocked, e_col, rtol=1e-11, atol=1e-12 * np.abs(e_col).max())
    np.testing.assert_allclose(np.linalg.eigvalsh(0.5 * (e_blocked + e_blocked.T)), np.linalg.eigvalsh(orc.E), rtol=1e-8)


def test_large_coarse_operator_blocked_cholesky(lib):
    """dimE = 312 (39 vectors x 8 subdomains, 64-column LOBPCG block): E goes through the blocked, threaded Cholesky
    and the transposed-factor back substitution instead of the small-matrix routine."""
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.9", "-geneo_cut", "39", "-ksp_type", "cg"] + TIGHT
    _, info = cases.compare_with_oracle(lib, 12, (2, 2, 2), 1, argv)
    assert info["dimE"] == 312


@pytest.mark.parametrize("style", ["PCGenEOSetup", "initGenEOPC"])
def test_reference_style_entry_points_one_subdomain_per_rank(lib, style):
    """The two entry styles of the reference (hdr/geneo_c.h:10 PCGenEOSetup after KSPSetOperators(MATIS);
    hdr/geneo.hpp:30 initGenEOPC): one subdomain = this rank's whole share.  On one rank the subdomain is the
    domain: no overlap, Z from the pencil (A, A) -> a single Nicolaides-type vector or eigenvalues 1; M^-1 b must be
    A^-1 b + Q b and the solve must return A^-1 b."""
    from geneo4petsc_amd import decomp
    from geneo4petsc_amd.pc import GenEOPC
    mesh = decomp.grid_mesh(size=6, dim=3)
    a = decomp.global_matrix(mesh).tocsr()
    n = a.shape[0]
    b = decomp.rhs_default(a)
    pc = GenEOPC(lib)
    pc.set_from_options(["-geneo_lvl", "ASM,0", "-ksp_type", "cg", "-ksp_rtol", "1e-10"])
    if style == "PCGenEOSetup":
        pc.setup_from_operators(n, np.arange(n), a, np.ones(n, dtype=np.int32), None, [np.zeros(0, dtype=np.int32)])
    else:
        pc.init(n, n, np.arange(n), a, None, None, None, np.arange(n), np.ones(n, dtype=np.uint32), [[]])
    pc.setup(b)
    x, its, rnorm, reason = pc.solve(b)
    assert reason.startswith("KSP_CONVERGED") and its <= 3            # the "preconditioner" is A^-1 itself
    np.testing.assert_allclose(x, np.arange(1.0, n + 1.0), rtol=1e-8)
    y = pc.apply(b)
    np.testing.assert_allclose(y, np.arange(1.0, n + 1.0), rtol=1e-8)
    pc.destroy()


def test_no_cut_more_eigenvalues_below_tau_than_one_block_holds(lib):
    """Row a7 (geneo.cpp:502-560, :713): without -geneo_cut the reference keeps EVERY eigenvalue below tau, whatever their
    number.  12^3 in 8 subdomains of 343 rows (> 192: the LOBPCG path, not the dense one), tau 0.6: 79 vectors per
    subdomain -- block growth 16 -> 32 -> 64, then deflated restarts (48 locked pairs per stage, the next block iterated
    in the B-orthogonal complement).  dimE, kept counts, eigenvalues (1e-10) and the GMRES count equal the oracle's."""
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.6", "-ksp_type", "gmres"] + TIGHT
    _, info = cases.compare_with_oracle(lib, 12, (2, 2, 2), 1, argv)
    assert info["dimE"] == 632 and info["realDimELoc"] == 632


def test_singular_neumann_matrices_get_null_pivot_fixing(lib):
    """--inpEps 0: the Neumann matrices of the four subdomains away from the Dirichlet face are singular (pure Neumann
    Laplacians).  The reference tells MUMPS to detect null pivots and fix them (tuneSolver, geneo.cpp:76-92: ICNTL(24),
    CNTL(5) = 1e20) for exactly this; here the coarsest blocks of the hierarchies pin them the same way
    (dense::cholesky_fix_null_pivots) -- without it the V-cycle amplifies the kernel component of every residual by
    1 / rounding and LOBPCG never converges.  What is compared: cases.check_singular_neumann_case."""
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", "gmres"] + TIGHT
    cases.check_singular_neumann_case(lib, 12, argv)


def test_memory_bounded_setup_groups_give_the_ungrouped_result(lib):
    """-geneo_eig_group_rows: eigensolve group by group (one subdomain per group, and three groups) == all at once, to the bit."""
    cases.check_grouped_eigensolve(lib, n=14)


def test_memory_bounded_setup_without_cut_and_with_the_chebyshev_eigensolver(lib):
    """the same with the deflated-restart path (no -geneo_cut: more pairs below tau than one block holds) inside a group"""
    mesh, dec, a, b = cases.grid_case(n=12, dim=3, parts=(2, 2, 2), overlap=1)
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.6", "-ksp_type", "gmres"] + TIGHT
    res = []
    for extra in ([], ["-geneo_eig_group_rows", "1000"]):
        pc = cases.run_pc(lib, mesh, dec, argv + extra, b)
        info = pc.info()
        res.append(([pc.eigenvalues(s) for s in range(8)], pc.E(), info["dimE"], info["eigGroups"]))
        pc.destroy()
    assert res[0][3] == 1 and res[1][3] == 4 and res[0][2] == res[1][2] == 632
    for s in range(8):
        assert np.array_equal(res[0][0][s], res[1][0][s])
    assert np.array_equal(res[0][1], res[1][1])


def test_local_solves_single_step_tail(lib, monkeypatch):
    """inner PCG of the local solves behind its long first chunk: single iterations (large subdomains; the row bound is
    lowered here) instead of pairs -- same outer count, same solution to the inner tolerance, never more inner iterations"""
    mesh, dec, a, b = cases.grid_case(n=16, dim=3, parts=(2, 2, 2), overlap=2)
    res = []
    for rows in ("1000000000", "1"):
        monkeypatch.setenv("GENEO_DLS1_SINGLE_STEP_ROWS", rows)
        pc = cases.run_pc(lib, mesh, dec, cases.bench_argv(), b)
        x, its, rnorm, reason = pc.solve(b)
        res.append((x, its, reason, pc.info()["dls1_iterations"]))
        pc.destroy()
    (x2, its2, r2, inner2), (x1, its1, r1, inner1) = res
    assert r1.startswith("KSP_CONVERGED") and its1 == its2
    assert inner1 <= inner2, (inner1, inner2)
    assert np.linalg.norm(x1 - x2) <= 1e-5 * np.linalg.norm(x2)


def test_eigensolve_coarse_start_gives_the_same_eigenpairs(lib):
    """-geneo_eig_coarse_start 1: LOBPCG on the Galerkin pencil of multigrid level 1 (A_c from the A_Neu hierarchy,
    B_c = P^T D A_Dir D P), its Ritz vectors prolonged as the start block of the fine iteration.  At a tight tolerance
    the result is the one of the random start: kept eigenvalues to 1e-8, kept counts, dimE, the PCG count."""
    off, on, worst = cases.check_coarse_start(lib, n=20, extra=["-geneo_cut", "8"])
    assert on["amg_levels"] >= 2
    print("coarse start: %d coarse + %d fine LOBPCG iterations (random start: %d), worst relative eigenvalue difference %.1e"
          % (on["eigCoarseIterations"], on["eig_iterations"], off["eig_iterations"], worst))


def test_eigensolve_coarse_start_stays_out_where_it_does_not_apply(lib):
    """small subdomains (default threshold), GenEO-2, and the block-growing path without -geneo_cut keep the random start"""
    mesh, dec, a, b = cases.grid_case(n=14, dim=3, parts=(2, 2, 2), overlap=2)
    for argv in (cases.bench_argv(),                                                        # 1 000 rows per subdomain < 750 000
                 cases.bench_argv(["-geneo_eig_coarse_start", "1", "-geneo_lvl", "SORAS,2", "-geneo_optim", "0.5"]),
                 ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-ksp_type", "gmres", "-geneo_eig_coarse_start", "1"] + TIGHT):
        pc = cases.run_pc(lib, mesh, dec, argv, b)
        assert pc.info()["eigCoarseIterations"] == 0, argv
        pc.destroy()


def test_setup_failure_with_one_oras_block_joins_the_hierarchy_thread(lib):
    """one subdomain per rank + ORAS + AMG local solves: the level-1 hierarchy thread reads the Robin matrix; an
    eigensolve that fails (here: a block narrower than -geneo_cut, refused before the first iteration) must come back as an error code with that thread joined
    and the matrix still alive -- it now lives in the pending object, not on setup()'s stack."""
    mesh, dec, a, b = cases.grid_case(n=10, dim=3, parts=(1, 1, 1), overlap=1)
    argv = ["-geneo_lvl", "ORAS,1", "-geneo_optim", "0.5", "-geneo_tau", "0.2", "-geneo_cut", "20", "-els2_eps_block", "16",
            "-ksp_type", "gmres"]
    from geneo4petsc_amd.pc import GenEOPC, GenEOError
    for _ in range(3):
        pc = GenEOPC(lib)
        pc.set_from_options(argv)
        pc.set_sizes(mesh.nbNode, 1)
        d = dec.domains[0]
        pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
        with pytest.raises(GenEOError, match="larger than the LOBPCG block"):
            pc.setup(b)
        pc.destroy()


def test_nicolaides_zero_window_option(lib):
    """-geneo_nicolaides_zero X: the window below which the smallest kept eigenvalue counts as the zero eigenvalue
    (geneo.cpp:896-899 tests min >= DBL_EPSILON, i.e. X = 1; the default is 100, see core.cpp).  Outside the window the two
    rules agree: on a regular problem X = 1 changes nothing.  On exactly singular Neumann matrices (--inpEps 0) the rule
    with X = 0 fires exactly on the subdomains whose computed zero eigenvalue came out >= 0 -- the reference's behaviour
    there is decided by the sign of a rounding error, which is why the default window exists."""
    mesh, dec, a, b = cases.grid_case(n=12, dim=3, parts=(2, 2, 2), overlap=1)
    argv = ["-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-geneo_cut", "8", "-ksp_type", "gmres"] + TIGHT
    ref = cases.run_pc(lib, mesh, dec, argv, b)
    lit = cases.run_pc(lib, mesh, dec, argv + ["-geneo_nicolaides_zero", "1"], b)
    assert ref.info()["nicolaidesLoc"] == lit.info()["nicolaidesLoc"] and np.array_equal(ref.E(), lit.E())
    ref.destroy()
    lit.destroy()
    mesh, dec, a, b = cases.grid_case(n=12, dim=3, parts=(2, 2, 2), overlap=1, inp_eps=0.0)
    wide = cases.run_pc(lib, mesh, dec, argv + ["-geneo_nicolaides_zero", "1e9"], b)
    assert wide.info()["nicolaidesLoc"] == 0
    lam0 = [float(np.min(wide.eigenvalues(s, candidates=True))) for s in range(8)]
    wide.destroy()
    none = cases.run_pc(lib, mesh, dec, argv + ["-geneo_nicolaides_zero", "0"], b)
    floating = [s for s in range(8) if abs(lam0[s]) < 1e-10]
    assert len(floating) == 4
    assert none.info()["nicolaidesLoc"] == sum(1 for s in floating if lam0[s] >= 0.0)
    none.destroy()
    from geneo4petsc_amd.pc import GenEOPC, GenEOError
    pc = GenEOPC(lib)
    with pytest.raises(GenEOError, match="nicolaides_zero"):
        pc.set_from_options(["-geneo_nicolaides_zero", "-3"])
    pc.destroy()
