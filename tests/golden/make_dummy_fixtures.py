#!/usr/bin/env python3
"""Extract the DATA of the reference's 84 golden logs (tst/dummy/*.ref) into one JSON fixture.

Run in the authoring container (the GPU box has no /root/reference):
    python tests/golden/make_dummy_fixtures.py

Each .ref is the stdout of one `mpirun -n 2 geneo4PETSc ...` run (tst/dummy/dummy.sh:61-66).
It pins, to PETSc print precision: the two per-rank MATIS local (Neumann) matrices (or the
assembled MPIAIJ matrix for -pc_type bjacobi), the right-hand side, the converged solution,
and the INFO lines (nnz, overlap, metis mode, PC name).  Only numbers/strings are kept.
The three tiny input data files of the same directory are carried along as data.
"""
import json
import os
import re
import sys

REF_DIR = "/root/reference/tst/dummy"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dummy_refs.json")

ROW = re.compile(r"^row (\d+):(.*)$")
ENT = re.compile(r"\((\d+), ([^)]+)\)")


def parse_ref(text):
    lines = text.splitlines()
    mats, cur = [], None
    mat_type = None
    b, x, info = [], [], []
    mode = None
    for ln in lines:
        s = ln.strip()
        if s.startswith("The matrix A is"):
            mode = "A"; continue
        if s.startswith("The vector B is"):
            mode = "B"; continue
        if s.startswith("The solution X is"):
            mode = "X"; continue
        if s.startswith("INFO:"):
            mode = None
            info.append(s)
            continue
        if mode == "A":
            if s.startswith("type:"):
                t = s.split()[1]
                if mat_type is None:
                    mat_type = t
                if t in ("seqaij", "mpiaij"):
                    cur = []
                    mats.append(cur)
                continue
            m = ROW.match(s)
            if m:
                ents = [[int(c), float(v)] for c, v in ENT.findall(m.group(2))]
                cur.append(ents)
        elif mode in ("B", "X"):
            try:
                v = float(s)
            except ValueError:
                continue
            (b if mode == "B" else x).append(v)
    return dict(mat_type=mat_type, mats=mats, b=b, x=x, info=info)


def parse_name(fn):
    # e.g. tridiag-pc=geneoASMH1-metis=dual-opt=overlap1.ref
    base = fn[:-4]
    parts = base.split("-")
    rec = dict(input=parts[0], pc=None, metis=None, opt="")
    for p in parts[1:]:
        k, v = p.split("=")
        rec[k] = v
    pc = rec["pc"]
    if pc == "bjacobi":
        rec["geneo_lvl"] = None
    else:
        m = re.match(r"geneo(ASM|SORAS)([HE]?\d)", pc)
        rec["geneo_lvl"] = m.group(1) + "," + m.group(2)
    rec["overlap"] = 1 if rec["opt"] == "overlap1" else 0
    rec["offload"] = rec["opt"] == "offload"
    return rec


def main():
    if not os.path.isdir(REF_DIR):
        sys.exit("reference not present; fixtures are generated in the authoring container only")
    out = dict(source="geneo4PETSc tst/dummy/*.ref (data only)", inputs={}, refs=[])
    for f in ("identity.inp", "tridiag.inp", "B.inp"):
        out["inputs"][f] = open(os.path.join(REF_DIR, f)).read()
    for fn in sorted(os.listdir(REF_DIR)):
        if not fn.endswith(".ref"):
            continue
        rec = parse_name(fn)
        rec["file"] = fn
        rec.update(parse_ref(open(os.path.join(REF_DIR, fn)).read()))
        # command-line facts from tst/dummy/dummy.sh:61-66
        rec["inpEps"] = 1.0 if rec["input"] == "tridiag" else 1e-4
        rec["geneo_cut"] = 10 if rec["input"] == "tridiag" else -1
        rec["use_b_file"] = rec["input"] == "identity"
        rec["ksp_rtol"] = 1e-12
        rec["ksp_atol"] = 1e-12
        out["refs"].append(rec)
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=0, separators=(",", ":"))
    print("wrote", OUT, len(out["refs"]), "refs")


if __name__ == "__main__":
    main()
