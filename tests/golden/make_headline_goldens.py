#!/usr/bin/env python3
"""Generates tests/golden/headline.json: the oracle at bench.py's own option set on the LARGE grids of the bench lines
(96^3 and 126^3 = BASELINE configs[1]), which the GPU tests and bench.py itself compare their dimE / kept counts / GMRES
and PCG iteration counts with.  Hours of CPU: run once in the build container, commit the JSON.

    python tests/golden/make_headline_goldens.py [--workers W] [--mode literal|exact] 96 126

Modes as in make_tight_goldens.py: "literal" = the reference's own call sequence (exact LU local solves, ARPACK
shift-invert AT -els2_eps_tol 1e-3, geneo.cpp:649-663); "exact" = certified-exact eigenpairs (the limit tol -> 0).
Per mode: PCG (the bench's Krylov method) with its spread under fixed 1e-14 .. 1e-8 operator perturbations, GMRES with
-geneo_lvl SRAS,1 and RAS,1 (configs[2] says RAS), dimE, kept vectors per subdomain, eigenvalues.

Every sparse LU of a subdomain uses a geometric nested-dissection ordering (oracle._PermutedLU: the same exact
factorisation as the default COLAMD one up to the elimination order, i.e. up to rounding -- checked against the committed
64^3 golden, same 25 PCG iterations -- at a fraction of the fill: eight COLAMD factors of a 286 k-row subdomain do not fit
this container's 62 GB)."""
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

OUT = os.path.join(HERE, "headline.json")
GMRES = dict(rtol=1e-5, atol=1e-50, max_it=10000, restart=100)
CG = dict(rtol=1e-5, atol=1e-50, max_it=10000)


def rec_of(res):
    return {"its": int(res.its), "reason": res.reason, "history": [float(v) for v in res.history],
            "x_norm": float(np.linalg.norm(res.x))}


def save(n, key, value):
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    data.setdefault(str(n), {})[key] = value
    json.dump(data, open(OUT, "w"), indent=1)


def cg_spread(orc, b):
    """PCG counts with the preconditioner replaced by S M^-1 S, S = I + delta diag(g) (make_tight_goldens.spread)"""
    out = {}
    for delta in (1e-14, 1e-12, 1e-10, 1e-8):
        counts = []
        for seed in (1, 2):
            d = 1.0 + delta * np.random.default_rng(seed).standard_normal(len(b))
            counts.append(int(go.ksp_cg(orc.matmult, lambda r: d * orc.apply(d * r), b, orc.x0, **CG).its))
        out["%.0e" % delta] = counts
    return out


def main():
    args = sys.argv[1:]
    workers, modes = 3, ["literal"]
    while args and args[0].startswith("--"):
        if args[0] == "--workers":
            workers = int(args[1])
        elif args[0] == "--mode":
            modes = args[1].split(",")
        args = args[2:]
    for n in [int(a) for a in args] or [96]:
        argv = cases.bench_argv()
        mesh, dec, a, b = cases.grid_case(n=n, dim=3, parts=(2, 2, 2), overlap=cases.BENCH_OVERLAP)
        for mode in modes:
            t0 = time.time()
            subs = [go.Subdomain(d.l2g, d.a_neu, d.mult, d.intersect) for d in dec.domains]
            orc = go.GenEOOracle(mesh.nbNode, subs, go.parse_options(argv))
            if mode == "literal":
                orc.dense_limit, orc.exact_eigs = 0, False
            else:
                orc.dense_limit, orc.exact_eigs = 4000, True
            orc.workers = workers
            orc.fill_perms = [go.nd_perm_for_grid_subdomain(d.l2g, n) for d in dec.domains]
            orc.setup(b)
            t_setup = time.time() - t0
            print(n, mode, "set-up %.0f s, dimE %d" % (t_setup, orc.dimE), flush=True)
            rec = {"argv": argv, "overlap": cases.BENCH_OVERLAP, "parts": [2, 2, 2], "n": n, "dimE": int(orc.dimE),
                   "realDimELoc": [int(v) for v in orc.realDimELoc], "nicolaides": int(sum(orc.nicolaidesLoc)),
                   "eigvals": [[float(v) for v in np.sort(e)] for e in orc.eigvals], "lu_order": "geometric nested dissection",
                   "oracle_setup_seconds": t_setup}
            rec["cg"] = rec_of(go.ksp_cg(orc.matmult, orc.apply, b, orc.x0, **CG))
            save(n, mode, rec)
            print(n, mode, "cg", rec["cg"]["its"], "%.0f s" % (time.time() - t0), flush=True)
            rec["gmres_sras"] = rec_of(go.ksp_gmres(orc.matmult, orc.apply, b, orc.x0, **GMRES))
            orc.o.lvl1SRAS = False
            rec["gmres_ras"] = rec_of(go.ksp_gmres(orc.matmult, orc.apply, b, orc.x0, **GMRES))
            orc.o.lvl1SRAS = True
            save(n, mode, rec)
            print(n, mode, "gmres sras", rec["gmres_sras"]["its"], "ras", rec["gmres_ras"]["its"], "%.0f s" % (time.time() - t0), flush=True)
            rec["cg_spread"] = cg_spread(orc, b)
            rec["oracle_seconds"] = time.time() - t0
            save(n, mode, rec)
            print(n, mode, "cg spread", rec["cg_spread"], "%.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
