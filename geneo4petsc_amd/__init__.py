"""geneo4petsc_amd -- MI355X-native GenEO preconditioner behind the geneo4PETSc PC-shell API.

  csrc/      hand-written HIP (gfx950) kernels + C++ core + C ABI  -> libgeneopc.so
  _lib.py    ctypes binding of include/geneo_c.h (fails loudly without the HIP library)
  pc.py      host mirror of the reference interface (createGenEOPC / initGenEOPC / setup / apply)
  decomp.py  host-side input producer (generators, decomposition, weighted assembly)
  comm.py    one-process-per-GPU transport callbacks over torch.distributed (RCCL / gloo)
"""
from .pc import GenEOPC, DeviceVector, Spmv, GenEOError, block_kernel  # noqa: F401
