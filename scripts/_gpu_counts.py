import sys, json, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import cases
from geneo4petsc_amd import _lib
lib = _lib.load()
G = json.load(open('tests/golden/benchopt.json'))
for n in (32, 48, 64):
    for extra in ([], ["-els2_eps_tol", "1e-6"], ["-els2_eps_tol", "1e-8", "-dls1_ksp_rtol", "1e-10"]):
        argv = cases.bench_argv() + extra
        mesh, dec, a, b = cases.grid_case(n=n, dim=3, parts=(2, 2, 2), overlap=2)
        pc = cases.run_pc(lib, mesh, dec, argv, b)
        x, its, rnorm, reason = pc.solve(b)
        info = pc.info()
        ev = max(np.max(np.abs(np.sort(pc.eigenvalues(s)) - np.array(G[str(n)]["eigvals"][s])) / np.array(G[str(n)]["eigvals"][s])) for s in range(8))
        h = np.array(pc.residual_history()); thr = 1e-5 * h[0]
        print(n, extra, "its", its, "oracle", G[str(n)]["its"], "eig_it", info["eig_iterations"], "max eig rel err %.1e" % ev, "tail", np.round(h[-4:] / thr, 2), flush=True)
        pc.destroy()
