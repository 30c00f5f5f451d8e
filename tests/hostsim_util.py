"""Test-side helpers: build + bind the host simulator of libgeneopc (tests/hostsim) and run a
decomposition through the package's host mirror (geneo4petsc_amd.pc.GenEOPC) with ANY bound library.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hostsim"))

from geneo4petsc_amd import _lib as L            # noqa: E402
from geneo4petsc_amd.pc import GenEOPC           # noqa: E402

_hostsim = None


def hostsim_lib():
    global _hostsim
    if _hostsim is None:
        import build as hs_build
        path = hs_build.build()
        _hostsim = L.bind(path)
        assert _hostsim.GeneoBackendName() == b"hostsim"
    return _hostsim


def make_pc(lib, n_global, domains, argv, with_dir=True):
    """domains: list of objects with gid, l2g, mult, a_neu, a_dir."""
    pc = GenEOPC(lib)
    pc.set_from_options(argv)
    pc.set_sizes(n_global, len(domains))
    for d in domains:
        pc.add_subdomain(d.gid, d.l2g, d.mult, d.a_neu, d.a_dir if with_dir else None)
    return pc
