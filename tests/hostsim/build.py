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
OUT_OMP = os.path.join(HERE, "libgeneopc_hostomp.so")


def build_omp(force=False):
    """The same sources with OpenMP over the rows of every block / vector loop (-DGENEO_HOST_OMP, -O3 -mavx2 -mfma): the
    CPU baseline of bench.py (`cpu_baseline.geneo_sample`): the library's own algorithm on the box's host cores.  Bench /
    test infrastructure only -- the package loads nothing but the HIP library."""
    return build(force, omp=True)


def build(force=False, omp=False):
    srcs = [os.path.join(CSRC, "core.cpp"), os.path.join(CSRC, "amg.cpp"), os.path.join(CSRC, "capi.cpp"),
            os.path.join(CSRC, "comm_rccl.cpp"), os.path.join(CSRC, "partition.cpp"), os.path.join(CSRC, "decompose.cpp"), os.path.join(CSRC, "driver_main.cpp"),
            os.path.join(HERE, "backend_host.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("core.h", "backend.h", "dense.h", "amg.h")] + \
        [os.path.join(ROOT, "include", "geneo_c.h")]
    out = OUT_OMP if omp else OUT
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    flags = ["-O3", "-mavx2", "-mfma", "-fopenmp", "-DGENEO_HOST_OMP"] if omp else ["-O2"]   # AVX2 + FMA: as the product's host code
    cmd = ["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I", CSRC, "-o", out] + srcs + ["-ldl"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True))
