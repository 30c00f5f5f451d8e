"""Builds the TEST-ONLY host simulator of libgeneopc (see backend_host.cpp).

It compiles the product's host orchestration (core.cpp, capi.cpp) against the serial backend so
that the CPU test-suite can check host logic (LOBPCG driver, batched-CG driver, E assembly,
apply modes, Krylov loop, C-ABI) against the oracle without a GPU.  Never imported by the package.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "geneo4petsc_amd", "csrc")
OUT = os.path.join(HERE, "libgeneopc_hostsim.so")


def build(force=False):
    srcs = [os.path.join(CSRC, "core.cpp"), os.path.join(CSRC, "amg.cpp"), os.path.join(CSRC, "capi.cpp"),
            os.path.join(CSRC, "comm_rccl.cpp"), os.path.join(CSRC, "partition.cpp"), os.path.join(CSRC, "decompose.cpp"),
            os.path.join(HERE, "backend_host.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("core.h", "backend.h", "dense.h", "amg.h")] + \
        [os.path.join(ROOT, "include", "geneo_c.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I", CSRC, "-o", OUT] + srcs + ["-ldl"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
