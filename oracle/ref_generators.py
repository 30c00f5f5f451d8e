"""TEST INFRASTRUCTURE ONLY -- the REAL reference input generators (tst/laplacian, tst/heat, tst/graph
`getInput` plugins) compiled from /root/reference by oracle/ref_build/Makefile into oracle/_ref/.
Used to validate the oracle's and the product's restatements of those generators."""
import ctypes as C
import os

from .driver_oracle import Mesh

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


def available():
    return all(os.path.exists(os.path.join(REF_DIR, f)) for f in
               ("liblaplacian.so", "libheat.so", "libgraph.so", "libgetinput_shim.so"))


def get_input(plugin: str, args: str) -> Mesh:
    """plugin in {"laplacian", "heat", "graph"}; args exactly as after --inpLibA on the reference CLI."""
    shim = C.CDLL(os.path.join(REF_DIR, "libgetinput_shim.so"))
    up = C.POINTER(C.c_uint)
    ne, nn, nptr, nidx, nm = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
    eptr, eidx, mats = up(), up(), C.POINTER(C.c_double)()
    rc = shim.ref_get_input(os.path.join(REF_DIR, "lib%s.so" % plugin).encode(), args.encode(), C.byref(ne),
                            C.byref(nn), C.byref(eptr), C.byref(nptr), C.byref(eidx), C.byref(nidx),
                            C.byref(mats), C.byref(nm))
    if rc:
        raise RuntimeError("reference getInput failed (%d)" % rc)
    m = Mesh(ne.value, nn.value, [eptr[i] for i in range(nptr.value)], [eidx[i] for i in range(nidx.value)], [])
    pos = 0
    for e in range(ne.value):
        k = m.elemPtr[e + 1] - m.elemPtr[e]
        m.elemSubMat.append([mats[pos + i] for i in range(k * k)])
        pos += k * k
    for p in (eptr, eidx, mats):
        shim.ref_free(p)
    return m
