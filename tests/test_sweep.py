"""The reference's own sweep matrix (tst/laplacian/laplacianRun.sh:25-65, tst/graph/graphRun.sh, tst/heat/heatRun.sh),
replayed through the library and compared with the oracle, combination by combination:

    20 GenEO PC strings  (ASM,0 | ASM,{1,H1,E1} x {-, --addOverlap 1, -geneo_offload} | SORAS,0 |
                          SORAS,{2,H2,E2} x {-, --addOverlap 1, -geneo_offload})
  x -geneo_optim {0, 0.02} (ORAS only)  x  (tau, gamma) in {(0.1, 8), (0.2, 12)}  x  tolerance {1e-4, 1e-5} (rtol = atol)
  GMRES without restart (-ksp_gmres_restart 1000 -ksp_max_it 1000), -els2_eps_tol 1e-2 -els2_eps_max_it 50, no -geneo_cut,
  inputs: laplacian --size 10 --kappa 2. lin | graph --size 10 --level 2 --noGround | heat --size 10 --kappa 2. quad
  --lbd 1. --dt 0.1 (all --inpEps 0.0001), 2 subdomains (the reference: mpirun -n 2), dual and nodal mesh partitions.

Metis is not installed: the two subdomains come from the package's k-way partitioner on the same dual / nodal graph the
reference hands to Metis (src/geneo4PETSc.cpp:381-445) -- an irregular ordering for the graph input, cf. configs[4].
Asserted per combination: converged reason, dimE, kept vectors per subdomain, Nicolaides count and the GMRES iteration
count, all EXACTLY the oracle's (GMRES: no tolerance on the count, tests/cases.py), the solution to the Krylov tolerance.
Without -geneo_cut every eigenvalue beyond the threshold enters Z (up to 80 per subdomain here): subdomains whose count
overflows the 64-column LOBPCG block take the library's dense path (core.cpp, DENSE_FALLBACK_ROWS).
"""
import numpy as np
import pytest

import cases
from geneo4petsc_amd import decomp
from oracle import geneo_oracle as go

PCS = ["ASM,0"] + ["%s%s" % (l, v) for l in ("ASM,1", "ASM,H1", "ASM,E1") for v in ("", " +ov", " +off")] + ["SORAS,0"] + \
      ["%s%s" % (l, v) for l in ("SORAS,2", "SORAS,H2", "SORAS,E2") for v in ("", " +ov", " +off")]
INPUTS = {
    "laplacian": lambda: decomp.grid_mesh(size=10, dim=3, kappa_max=2.0, interp="lin", inp_eps=1e-4),
    "graph": lambda: decomp.graph_mesh(size=10, level=2, no_ground=True, inp_eps=1e-4),
    "heat": lambda: decomp.grid_mesh(size=10, dim=3, kappa_max=2.0, interp="quad", heat=True, lbd=1.0, dt=0.1, inp_eps=1e-4),
}
_cache = {}


def problem(inp, dual, overlap):
    key = (inp, dual, overlap)
    if key not in _cache:
        if (inp, dual) not in _cache:
            mesh = INPUTS[inp]()
            ep, npart = decomp.partition_mesh(mesh, 2, dual)
            a = decomp.global_matrix(mesh)
            _cache[(inp, dual)] = (mesh, ep, npart, a, decomp.rhs_default(a))
        mesh, ep, npart, a, b = _cache[(inp, dual)]
        _cache[key] = (mesh, decomp.decompose(mesh, 2, ep, npart, dual, overlap), a, b)
    return _cache[key]


def combos(pc):
    lvl = pc.split(" ")[0]
    l2 = lvl.split(",")[1]
    taus = [(0.1, 8.0), (0.2, 12.0)] if l2 != "0" else [(0.1, 8.0)]
    optims = [0.0, 0.02] if "ORAS" in lvl else [0.0]
    return [(t, g, o) for (t, g) in taus for o in optims]


def run_sweep(lib, inp, dual, pc):
    lvl = pc.split(" ")[0]
    overlap = 1 if "+ov" in pc else 0
    mesh, dec, a, b = problem(inp, dual, overlap)
    checked = 0
    for tau, gamma, optim in combos(pc):
        argv = ["-geneo_lvl", lvl, "-geneo_tau", str(tau), "-geneo_gamma", str(gamma), "-geneo_optim", "%.2f" % optim,
                "-ksp_type", "gmres", "-ksp_max_it", "1000", "-ksp_gmres_restart", "1000", "-els2_eps_tol", "1e-2",
                "-els2_eps_max_it", "50"] + (["-geneo_offload"] if "+off" in pc else [])
        lib_pc = cases.run_pc(lib, mesh, dec, argv + ["-ksp_rtol", "1e-4", "-ksp_atol", "1e-4"], b, with_intersect=True)
        orc = cases.oracle_for(mesh, dec, argv, b)
        info = lib_pc.info()
        tag = "%s %s %s tau %.1f gamma %.0f optim %.2f" % (inp, "dual" if dual else "nodal", pc, tau, gamma, optim)
        if orc.o.lvl2:
            assert [int(v) for v in lib_pc.local_dims()] == list(orc.realDimELoc), tag
            assert info["dimE"] == orc.dimE and info["nicolaidesLoc"] == sum(orc.nicolaidesLoc), tag
        for tol in ("1e-4", "1e-5"):
            lib_pc.set_option("-ksp_rtol", tol)
            lib_pc.set_option("-ksp_atol", tol)
            x, its, rnorm, reason = lib_pc.solve(b)
            res = go.solve(orc, b, "gmres", rtol=float(tol), atol=float(tol), max_it=1000, restart=1000)
            assert reason == res.reason, (tag, tol, reason, res.reason)
            assert its == res.its, "%s tol %s: GMRES iterations %d vs oracle %d" % (tag, tol, its, res.its)
            assert np.linalg.norm(x - res.x) <= 50 * float(tol) * np.linalg.norm(res.x), (tag, tol)
            checked += 1
        lib_pc.destroy()
    return checked


@pytest.mark.gpu
@pytest.mark.parametrize("pc", PCS)
@pytest.mark.parametrize("dual", [False, True], ids=["nodal", "dual"])
@pytest.mark.parametrize("inp", list(INPUTS))
def test_reference_sweep_on_gpu(inp, dual, pc):
    from geneo4petsc_amd import _lib
    assert run_sweep(_lib.load(), inp, dual, pc) >= 2


@pytest.mark.parametrize("inp,dual,pc", [("laplacian", False, "ASM,1"), ("graph", False, "SORAS,2"), ("heat", True, "SORAS,0")])
def test_reference_sweep_host_logic(inp, dual, pc):
    """A slice of the same matrix through the host logic on the test-only serial backend (CPU suite)."""
    import hostsim_util as hu
    assert run_sweep(hu.hostsim_lib(), inp, dual, pc) >= 2
