"""Counterpart of the reference CLI driver (src/geneo4PETSc.cpp main():1569) for the MI355X path.

    python -m geneo4petsc_amd.driver --inpLibA laplacian#--size#10#--dim#3 --np 8 --parts 2,2,2 \
           --metisNodal --addOverlap 1 -geneo_lvl ASM,1 -ksp_type cg --timing

Same flags where they make sense without MPI/Metis (driver:1396-1495): --inpFileA, --inpLibA (the three
reference generators by name: laplacian | heat | graph, arguments '#'-separated as on the reference CLI),
--inpFileB, --inpEps, --metisDual / --metisNodal, --addOverlap, --verbose, --timing, --shortRes, --cmdLine;
PETSc-style options (-geneo_*, -ksp_*, -els2_*, -dls1_*) are forwarded to the PC.  `--np N` stands for
`mpirun -n N` (number of subdomains); the partition comes from `--parts px,py,pz` (structured block
splitter), `--partFile` (one part id per line: a Metis .part file) or, by default, the built-in k-way
partitioner of decomp.partition_mesh on the dual (--metisDual) or nodal (--metisNodal) graph.

Output: the reference's INFO: / TIME: lines in the shapes tst/plot.py:57-116 parses
(printIterativeGlobalSolveParameters/Results/Timing, driver:898-1231).
"""
import sys
import time

import numpy as np

from . import decomp
from .pc import GenEOPC, DeviceVector


def parse_cli(argv):
    o = dict(inpFileA="", inpLibA="", inpFileB="", inpEps=1e-4, metisDual=True, addOverlap=0, verbose=0,
             timing=False, shortRes=False, cmdLine=False, np=1, parts=None, partFile="", pc_args=[])
    i = 0
    while i < len(argv):
        a = argv[i]
        nxt = argv[i + 1] if i + 1 < len(argv) else None
        if a in ("--inpFileA", "--inpLibA", "--inpFileB", "--partFile"):
            o[a[2:]] = nxt; i += 2
        elif a == "--inpEps":
            o["inpEps"] = float(nxt); i += 2
        elif a == "--metisDual":
            o["metisDual"] = True; i += 1
        elif a == "--metisNodal":
            o["metisDual"] = False; i += 1
        elif a == "--addOverlap":
            o["addOverlap"] = int(nxt); i += 2
        elif a == "--verbose":
            o["verbose"] = int(nxt); i += 2
        elif a == "--np":
            o["np"] = int(nxt); i += 2
        elif a == "--parts":
            o["parts"] = tuple(int(t) for t in nxt.split(",")); i += 2
        elif a in ("--timing", "--shortRes", "--cmdLine"):
            o[a[2:]] = True; i += 1
        elif a == "--debug":
            i += 2 if nxt and not nxt.startswith("-") else 1
        else:
            o["pc_args"].append(a); i += 1
    return o


def load_mesh(o, lib=None):
    """--inpFileA / --inpLibA (driver:144-194, :75-96).  --inpLibA takes either the path of a `getInput` plugin
    built for the reference driver (loaded unchanged through GeneoGetLibInput) or the bare name of one of the
    three reference generators restated in decomp.py (laplacian | heat | graph)."""
    import os
    if o["inpFileA"]:
        return decomp.read_input_text(open(o["inpFileA"]).read(), o["inpEps"]), None
    name, _, rest = o["inpLibA"].partition("#")
    if os.path.isfile(name):
        from . import _lib
        return decomp.plugin_mesh(lib if lib is not None else _lib.load(), name, rest), None
    name = name.split("/")[-1].replace("lib", "").replace(".so", "")
    tok = rest.replace("#", " ").split()
    kw = {}
    j = 0
    while j < len(tok):
        t = tok[j]
        if t in ("--size", "--level", "--weakScaling", "--dim"):
            kw[{"--size": "size", "--level": "level", "--weakScaling": "weak", "--dim": "dim"}[t]] = int(tok[j + 1]); j += 2
        elif t in ("--inpEps", "--lbd", "--dt"):
            kw[{"--inpEps": "inp_eps", "--lbd": "lbd", "--dt": "dt"}[t]] = float(tok[j + 1]); j += 2
        elif t == "--kappa":
            kw["kappa_max"] = float(tok[j + 1]); kw["interp"] = tok[j + 2]; j += 3
        elif t == "--noGround":
            kw["no_ground"] = True; j += 1
        else:
            j += 1
    if name == "graph":
        return decomp.graph_mesh(**kw), None
    if name == "heat":
        kw["heat"] = True
    kw.setdefault("dim", 3)
    mesh = decomp.grid_mesh(**kw)
    return mesh, (decomp.grid_size(kw.get("size", 4), kw.get("weak", 1), kw["dim"]), kw["dim"])


def make_partition(o, mesh, grid):
    nb = o["np"]
    if o["partFile"]:
        part = np.loadtxt(o["partFile"], dtype=np.int64).reshape(-1)
        return (part, None) if o["metisDual"] else (None, part)
    if not o["metisDual"] and grid is not None and o["parts"] is not None:
        return None, decomp.structured_node_partition(grid[0], grid[1], o["parts"])
    # Metis is not available offline: built-in k-way partitioner on the same dual / nodal graph (driver:381-445)
    return decomp.partition_mesh(mesh, nb, o["metisDual"])


def info_lines(o, mesh, nnz, pc, info, ksp, its, rnorm, reason, res_rel, nb_part):
    """printIterativeGlobalSolveParameters / Results (driver:898-1095)."""
    opt = pc.options()
    out = ["INFO: nb DOFs %d, nb elements %d, nnz coefs %d, nb partitions %d, overlap %d, metis %s"
           % (mesh.nbNode, mesh.nbElem, nnz, nb_part, o["addOverlap"], "dual" if o["metisDual"] else "nodal"),
           "INFO: %s ksp, eps rel %.1e, eps abs %.1e, max iterations %d" % (ksp["type"], ksp["rtol"], ksp["atol"],
                                                                             ksp["max_it"])]
    line = "INFO: %s pc" % pc.name
    if "ORAS" in pc.name:
        line += ", optim %.2f" % opt["optim"]
    if opt["effHybrid"]:
        line += ", initial guess"
    line += ", L1 %s %s" % ("pcg-" + opt["dls1_pc"], "proj-fine-space" if opt["hybrid"] else "no-proj-fine-space")
    if opt["lvl2"]:
        line += ", tau %.2f" % opt["tau"]
        if opt["lvl2"] >= 2:
            line += ", gamma %.2f" % opt["gamma"]
        if opt["offload"]:
            line += ", offload"
        line += ", L2 lobpcg cholesky"
    out.append(line)
    if not o["shortRes"]:
        if opt["lvl2"]:
            dims = pc.local_dims()
            out.append("INFO: setup - estim dimE %i (local: min %i, max %i), , real dimE %i (local: min %i, max %i)"
                       ", nicolaides %i" % (info["estimDimELoc"], int(dims.min()), int(dims.max()), info["dimE"],
                                            int(dims.min()), int(dims.max()), info["nicolaidesLoc"]))
        else:
            out.append("INFO: setup - none")
    conv = "converged" if reason.startswith("KSP_CONVERGED") else "diverged"
    if o["shortRes"]:
        out.append("INFO: solve - " + conv)
    else:
        out.append("INFO: solve - %s (%s), %d iteration(s), residual norm %.10f, || AX - B || / || B || %.10f"
                   % (conv, reason, its, rnorm, res_rel))
    return out


def time_lines(t_read, t_part, t_create, info, opt):
    """printIterativeGlobalSolveTiming (driver:1097-1231)."""
    out = ["TIME: read input %.5f s, part / decomp %.5f s, create A %.5f s, solver set up %.5f s, "
           "solver iterations %.5f s, solve %.5f s" % (t_read, t_part, t_create, info["setupTime"], info["solveTime"],
                                                        info["solveTime"] + info["setupTime"]),
           "      L1       setup: Minv %.5f s" % info["lvl1SetupMinvTimeLoc"]]
    if opt["lvl2"]:
        out.append("      L2       setup: eigen solve %.5f s, Z %.5f s, E %.5f s"
                   % (info["lvl2SetupEigTimeLoc"], info["lvl2SetupZTimeLoc"], info["lvl2SetupETimeLoc"]))
    out.append("      L1       solve: apply %.5f s - scatter %.5f s, Minv %.5f s, gather %.5f s"
               % (info["lvl1ApplyTimeLoc"], info["lvl1ApplyScatterTimeLoc"], info["lvl1ApplyMinvTimeLoc"],
                  info["lvl1ApplyGatherTimeLoc"]))
    if opt["lvl2"]:
        out.append("      L2       solve: apply %.5f s - Zt %.5f s, Einv %.5f s, Z %.5f s"
                   % (info["lvl2ApplyTimeLoc"], info["lvl2ApplyZtTimeLoc"], info["lvl2ApplyEinvTimeLoc"],
                      info["lvl2ApplyZTimeLoc"]))
    return out


def run(argv, lib=None, out=None):
    """Returns (lines, x).  `lib`: an already bound library (tests); default = the HIP product library."""
    out = out if out is not None else sys.stdout
    o = parse_cli(argv)
    t0 = time.perf_counter()
    mesh, grid = load_mesh(o, lib)
    t_read = time.perf_counter() - t0
    t0 = time.perf_counter()
    ep, npart = make_partition(o, mesh, grid)
    dec = decomp.decompose(mesh, o["np"], ep, npart, o["metisDual"], o["addOverlap"])
    t_part = time.perf_counter() - t0
    t0 = time.perf_counter()
    a = decomp.global_matrix(mesh)
    b = decomp.read_b_text(open(o["inpFileB"]).read(), mesh.nbNode) if o["inpFileB"] else decomp.rhs_default(a)
    nnz = sum(int(d.a_neu.indptr[-1]) for d in dec.domains)
    pc = GenEOPC(lib)
    pc.set_from_options(o["pc_args"])
    pc.set_sizes(mesh.nbNode, o["np"])
    for d in dec.domains:
        pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir)
    t_create = time.perf_counter() - t0
    pc.setup(b)
    x, its, rnorm, reason = pc.solve(b)
    info = pc.info()
    res_rel = float(np.linalg.norm(a @ x - b) / np.linalg.norm(b))
    opt = pc.options()
    ksp = dict(type=opt["ksp_type"], rtol=opt["ksp_rtol"], atol=opt["ksp_atol"], max_it=opt["ksp_max_it"])
    lines = []
    if o["cmdLine"]:
        lines.append("CMD: " + " ".join(argv))
    if o["verbose"] >= 1:
        lines += ["The solution X is:"] + ["%g" % v for v in x] + [""]
    lines += info_lines(o, mesh, nnz, pc, info, ksp, its, rnorm, reason, res_rel, o["np"])
    if o["timing"]:
        lines += [""] + time_lines(t_read, t_part, t_create, info, opt)
    for ln in lines:
        print(ln, file=out)
    pc.destroy()
    return lines, x


if __name__ == "__main__":
    run(sys.argv[1:])
