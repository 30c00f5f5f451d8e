import sys, numpy as np, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import cases
from geneo4petsc_amd import _lib
lib=_lib.load()
n=int(sys.argv[1])
G=json.load(open('/root/repo/tests/golden/benchopt.json'))[str(n)]
mesh, dec, a, b = cases.grid_case(n=n, dim=3, parts=(2,2,2), overlap=2)
for extra in ([], ["-els2_eps_tol","5e-4"], ["-els2_eps_tol","2e-4"]):
    pc = cases.run_pc(lib, mesh, dec, cases.bench_argv(extra), b)
    x, its, rn, reason = pc.solve(b)
    h = np.array(pc.residual_history()); thr = 1e-5*h[0]
    print(n, extra, "its", its, "eig its", pc.info()["eig_iterations"], "tail/thr", np.round(h[-5:]/thr,3), "oracle literal", G["literal"]["its"], "exact", G["exact_eigs"]["its"], flush=True)
    pc.destroy()
