"""Builds the TEST-ONLY librccl stand-in (rccl_standin.cpp: RCCL's point-to-point / all-reduce API between processes of
one host over shared memory, host pointers).  Selected by GENEO_RCCL_LIBRARY in tests/test_rccl_two_peers.py only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rccl_standin.cpp")
OUT = os.path.join(HERE, "librccl_standin.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", OUT, SRC, "-lrt"])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
