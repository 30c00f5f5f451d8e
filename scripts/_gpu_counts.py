import sys, json, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import cases
from geneo4petsc_amd import _lib
lib = _lib.load()
G = json.load(open('tests/golden/benchopt.json'))
for n in (24, 28, 32, 48, 64):
    for extra in ([], ["-els2_eps_tol", "5e-4"], ["-els2_eps_tol", "2.5e-4"]):
        argv = cases.bench_argv() + extra
        mesh, dec, a, b = cases.grid_case(n=n, dim=3, parts=(2, 2, 2), overlap=2)
        pc = cases.run_pc(lib, mesh, dec, argv, b)
        x, its, rnorm, reason = pc.solve(b)
        info = pc.info()
        h = np.array(pc.residual_history()); thr = 1e-5 * h[0]
        g = G.get(str(n), {})
        print(n, extra, "its", its, "literal", g.get("literal", {}).get("its"), g.get("literal_counts_by_seed"), "exact", g.get("exact_eigs", {}).get("its"),
              "eig_it", info["eig_iterations"], "tail", np.round(h[-4:] / thr, 2), flush=True)
        pc.destroy()
