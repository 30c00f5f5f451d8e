"""Shared helpers for the reference's tst/dummy golden logs (tests/golden/dummy_refs.json).

Metis 5.1.0 is not available offline, and the goldens embed its partition only implicitly
(through the printed per-rank local matrices).  The 8-node chain is small enough to recover
the partition by exhaustive search: the first element/node partition whose decomposition and
weighted assembly reproduce the printed matrices of BOTH ranks, in rank order, is the fixture.
"""
import itertools
import json
import os
from functools import lru_cache

import numpy as np

from oracle import driver_oracle as drv

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dummy_refs.json")


@lru_cache(maxsize=1)
def load():
    with open(GOLDEN) as fh:
        return json.load(fh)


def mesh_for(rec):
    d = load()
    return drv.read_input_text(d["inputs"][rec["input"] + ".inp"], rec["inpEps"])


def rows_of(a, tol=0.0):
    """CSR -> [[(col, val), ...] per row] without explicit zeros (PETSc ASCII_COMMON view)."""
    out = []
    a = a.tocsr()
    a.sort_indices()
    for i in range(a.shape[0]):
        s, t = a.indptr[i], a.indptr[i + 1]
        out.append([[int(c), float(v)] for c, v in zip(a.indices[s:t], a.data[s:t]) if abs(v) > tol])
    return out


def same_rows(got, want, rtol=1e-5):
    if len(got) != len(want):
        return False
    for rg, rw in zip(got, want):
        if len(rg) != len(rw):
            return False
        for (cg, vg), (cw, vw) in zip(rg, rw):
            if cg != cw or abs(vg - vw) > rtol * max(1.0, abs(vw)):
                return False
    return True


@lru_cache(maxsize=None)
def find_partition(input_name, inp_eps, metis, overlap, want_json):
    """Exhaustive search over 2-way partitions; returns (elem_part|None, node_part|None)."""
    d = load()
    mesh = drv.read_input_text(d["inputs"][input_name + ".inp"], inp_eps)
    want = json.loads(want_json)
    dual = metis == "dual"
    nitems = mesh.nbElem if dual else mesh.nbNode
    for bits in itertools.product((0, 1), repeat=nitems):
        if len(set(bits)) < 2:
            continue
        ep = list(bits) if dual else None
        npart = None if dual else list(bits)
        dec = drv.decompose(mesh, 2, ep, npart, dual, overlap)
        ok = True
        for p in range(2):
            a = drv.assemble_local(mesh, dec, p)
            if not same_rows(rows_of(a), want[p]):
                ok = False
                break
        if ok:
            return (tuple(ep) if ep else None, tuple(npart) if npart else None)
    return None


def geneo_refs():
    return [r for r in load()["refs"] if r["mat_type"] == "is"]


def bjacobi_refs():
    return [r for r in load()["refs"] if r["mat_type"] == "mpiaij"]


def partition_for(rec):
    """Partition fixture for one golden (the MATIS goldens of the same input/metis/overlap pin it)."""
    donor = rec
    if rec["mat_type"] != "is":
        for r in geneo_refs():
            if (r["input"], r["metis"], r["overlap"]) == (rec["input"], rec["metis"], rec["overlap"]):
                donor = r
                break
    return find_partition(donor["input"], donor["inpEps"], donor["metis"], donor["overlap"],
                          json.dumps(donor["mats"]))


def rhs_for(rec, a_global):
    d = load()
    n = a_global.shape[0]
    if rec["use_b_file"]:
        return drv.read_b_text(d["inputs"]["B.inp"], n)
    return a_global @ np.arange(1.0, n + 1.0)      # driver:820-831
