#!/usr/bin/env python3
"""Generates tests/golden/benchopt_tight.json: the oracle at bench.py's option set in the limit where the coarse space is
unique (certified-exact eigenpairs = the limit -els2_eps_tol -> 0, exact LU local solves), plus the GMRES / RAS variants
BASELINE configs[2] names, on the grids of tests/test_bench_options.py (32^3, 48^3, 64^3: minutes of CPU each).

Per grid, from ONE exact set-up (RAS / SRAS only differ in the application, geneo.cpp:1991-2002):
  exact.cg            PCG, rtol 1e-5 (the bench's Krylov method): count + residual history
  exact.cg_spread     the SAME PCG with the preconditioner perturbed as a fixed operator, S M^-1 S with S = I + delta diag(g)
                      (delta = 1e-14 .. 1e-8, four seeds each): the spread of the oracle's OWN count under perturbations
                      no two implementations can avoid (summation orders, exact LU against an iterative local solve,
                      eigenvectors converged to a tolerance).  Once a Ritz value of the Krylov process has converged to
                      delta, the two runs are different Lanczos processes (Paige): the tail of the history is no longer
                      the same sequence, and the count moves whenever the residual crosses the threshold there.
  exact.gmres_*_spread  the same for GMRES
  exact.gmres_sras    GMRES (restart 100: no restart inside the solve) with the same preconditioner
  exact.gmres_ras     GMRES with -geneo_lvl RAS,1 (configs[2] says RAS; non-symmetric => GMRES, laplacianRun.sh:52)
and from one reference-literal set-up (ARPACK shift-invert AT -els2_eps_tol 1e-3, geneo.cpp:649-663):
  literal.gmres_ras / literal.gmres_sras

    python tests/golden/make_tight_goldens.py 32 48 64
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases                                  # noqa: E402
from oracle import geneo_oracle as go         # noqa: E402

OUT = os.path.join(HERE, "benchopt_tight.json")
GMRES = dict(rtol=1e-5, atol=1e-50, max_it=10000, restart=100)
CG = dict(rtol=1e-5, atol=1e-50, max_it=10000)


def rec_of(res):
    return {"its": int(res.its), "reason": res.reason, "history": [float(v) for v in res.history],
            "x_norm": float(np.linalg.norm(res.x))}


def save(n, key, value):
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    data.setdefault(str(n), {})[key] = value
    json.dump(data, open(OUT, "w"), indent=1)


def spread(orc, b, fn, kw):
    """Counts of the same solve with the preconditioner replaced by S M^-1 S, S = I + delta diag(g), g standard normal:
    a FIXED symmetric relative perturbation of the operator, as two correct implementations differ from each other
    (delta ~ 1e-12: two backward-stable LU factorisations of a matrix with condition 1e4; 1e-10 / 1e-8: eigenvectors
    converged to that tolerance)."""
    out = {}
    for delta in (1e-14, 1e-12, 1e-10, 1e-8):
        counts = []
        for seed in (1, 2, 3, 4):
            d = 1.0 + delta * np.random.default_rng(seed).standard_normal(len(b))
            counts.append(int(fn(orc.matmult, lambda r: d * orc.apply(d * r), b, orc.x0, **kw).its))
        out["%.0e" % delta] = counts
    return out


def variants(orc, b, with_spread):
    out = {}
    out["cg"] = rec_of(go.ksp_cg(orc.matmult, orc.apply, b, orc.x0, **CG))
    out["gmres_sras"] = rec_of(go.ksp_gmres(orc.matmult, orc.apply, b, orc.x0, **GMRES))
    if with_spread:
        out["cg_spread"] = spread(orc, b, go.ksp_cg, CG)
        out["gmres_sras_spread"] = spread(orc, b, go.ksp_gmres, GMRES)
    orc.o.lvl1SRAS = False                    # RAS,1: same set-up, [D] only on the way in (geneo.cpp:1991-2002)
    out["gmres_ras"] = rec_of(go.ksp_gmres(orc.matmult, orc.apply, b, orc.x0, **GMRES))
    if with_spread:
        out["gmres_ras_spread"] = spread(orc, b, go.ksp_gmres, GMRES)
    orc.o.lvl1SRAS = True
    return out


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [32]
    for n in sizes:
        argv = cases.bench_argv()
        mesh, dec, a, b = cases.grid_case(n=n, dim=3, parts=(2, 2, 2), overlap=cases.BENCH_OVERLAP)
        for mode in ("exact", "literal"):
            t0 = time.time()
            orc = cases.oracle_for(mesh, dec, argv, b, literal=(mode == "literal"))
            rec = variants(orc, b, with_spread=(mode == "exact"))
            rec.update({"argv": argv, "overlap": cases.BENCH_OVERLAP, "parts": [2, 2, 2], "n": n, "dimE": int(orc.dimE),
                        "realDimELoc": [int(v) for v in orc.realDimELoc],
                        "eigvals": [[float(v) for v in np.sort(e)] for e in orc.eigvals],
                        "oracle_seconds": time.time() - t0})
            save(n, mode, rec)
            print(n, mode, {k: (v["its"] if isinstance(v, dict) and "its" in v else v) for k, v in rec.items()
                            if k in ("cg", "cg_spread", "gmres_sras", "gmres_sras_spread", "gmres_ras", "gmres_ras_spread")}, "%.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
