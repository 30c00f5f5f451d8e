"""One-process-per-GPU transport for libgeneopc over torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).  PyTorch is plumbing only: it owns the staging
buffers and moves them; every numerical kernel runs in the HIP library.

The GenEO hot path has exactly two exchange steps:
  * halo forward / reverse (R_i and R_i^T across ranks): neighbour data, one all_to_all_single with
    the precomputed split sizes -- only neighbours have non-zero splits, so over RCCL this becomes
    point-to-point sends on the direct xGMI links;
  * tiny all-reduce (Z^T x of dimE doubles, Krylov dots, subdomain sizes): latency-bound.
"""
import numpy as np


class TorchComm:
    WIDTH = 32     # vectors per halo exchange the buffers are sized for (blocked assembly of E)

    def __init__(self, plan, device, red_capacity=1 << 16):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.plan = plan
        self.send_counts = [int(c) for c in plan.send_counts]
        self.recv_counts = [int(c) for c in plan.recv_counts]
        self.nsend, self.nrecv = sum(self.send_counts), sum(self.recv_counts)
        cap = max(1, self.nsend, self.nrecv) * self.WIDTH
        self.send = torch.zeros(cap, dtype=torch.float64, device=device)
        self.recv = torch.zeros(cap, dtype=torch.float64, device=device)
        self.red = torch.zeros(red_capacity, dtype=torch.float64, device=device)
        self.red_capacity = red_capacity
        self.error = None

    def exchange(self, user, flag):
        try:
            reverse, w = flag & 1, max(1, flag >> 1)      # w vectors per entry, entry-major
            sc = self.send_counts if w == 1 else [c * w for c in self.send_counts]
            rc = self.recv_counts if w == 1 else [c * w for c in self.recv_counts]
            if reverse:   # halo contributions travel back to their owners
                self.dist.all_to_all_single(self.recv[:self.nsend * w], self.send[:self.nrecv * w],
                                            output_split_sizes=sc, input_split_sizes=rc)
            else:         # owners send the values of their DOFs to every rank that overlaps them
                self.dist.all_to_all_single(self.recv[:self.nrecv * w], self.send[:self.nsend * w],
                                            output_split_sizes=rc, input_split_sizes=sc)
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            self.error = e
            return 1

    def allreduce(self, user, n):
        try:
            self.dist.all_reduce(self.red[:n])
            return 0
        except Exception as e:
            self.error = e
            return 1

    def attach(self, pc):
        p = self.plan
        pc.set_comm(p.rank, p.size, p.owned, p.halo_gid, p.recv_counts, p.send_counts, p.send_idx,
                    self.exchange, self.allreduce, self.send.data_ptr(), self.recv.data_ptr(),
                    self.red.data_ptr(), self.red_capacity)
        pc.set_comm_width(self.WIDTH)


class StagedComm(TorchComm):
    """Same two exchange steps through host staging: the library's device buffers are copied to CPU tensors,
    moved with a CPU backend (gloo) and copied back.  For clusters whose transport cannot take device pointers,
    and for rehearsing the N > 1 path of the HIP library with several ranks on ONE GPU (tests/test_gpu_multirank.py)."""

    def __init__(self, plan, lib, red_capacity=1 << 16):
        super().__init__(plan, "cpu", red_capacity)
        from .pc import DeviceVector
        self.lib = lib
        cap = max(1, self.nsend, self.nrecv) * self.WIDTH
        self.d_send, self.d_recv = DeviceVector(lib, cap), DeviceVector(lib, cap)
        self.d_red = DeviceVector(lib, red_capacity)

    def _down(self, dev, host, n):
        if n and self.lib.GeneoD2H(host.data_ptr(), dev.ptr, n * 8):
            raise RuntimeError("D2H failed")

    def _up(self, dev, host, n):
        if n and self.lib.GeneoH2D(dev.ptr, host.data_ptr(), n * 8):
            raise RuntimeError("H2D failed")

    def exchange(self, user, flag):
        try:
            reverse, w = flag & 1, max(1, flag >> 1)
            nout, nin = (self.nrecv, self.nsend) if reverse else (self.nsend, self.nrecv)
            self._down(self.d_send, self.send, nout * w)
            rc = super().exchange(user, flag)
            self._up(self.d_recv, self.recv, nin * w)
            return rc
        except Exception as e:
            self.error = e
            return 1

    def allreduce(self, user, n):
        try:
            self._down(self.d_red, self.red, n)
            rc = super().allreduce(user, n)
            self._up(self.d_red, self.red, n)
            return rc
        except Exception as e:
            self.error = e
            return 1

    def attach(self, pc):
        p = self.plan
        pc.set_comm(p.rank, p.size, p.owned, p.halo_gid, p.recv_counts, p.send_counts, p.send_idx,
                    self.exchange, self.allreduce, self.d_send.ptr, self.d_recv.ptr, self.d_red.ptr, self.red_capacity)
        pc.set_comm_width(self.WIDTH)


class RcclComm:
    """The library's own C++ transport over RCCL / xGMI (csrc/comm_rccl.cpp): ncclSend / ncclRecv groups for the halo,
    ncclAllReduce for the reductions, on the library's stream -- no Python callback in the data path.  Python only
    bootstraps: rank 0 makes the 128-byte unique id, torch.distributed (any backend) broadcasts it."""
    WIDTH = 32

    def __init__(self, plan, lib, dist=None, device=None):
        import ctypes as C
        self.lib, self.plan, self.error = lib, plan, None
        ida = C.create_string_buffer(128)
        err = None
        if plan.rank == 0 and lib.GeneoRcclUniqueId(ida):
            err = lib.GeneoRcclGetError().decode()
        self.h = C.c_void_p()
        if plan.size > 1:
            # Every step is agreed on by all ranks before anyone raises: a rank that fails alone (RCCL not loadable,
            # communicator not created) must not leave the others waiting in a collective -- the caller can then fall
            # back to another transport on ALL ranks (bench.py does).
            import torch
            dev = device if device is not None else "cpu"
            t = torch.tensor([0 if err else 1] + list(ida.raw), dtype=torch.uint8, device=dev)
            dist.broadcast(t, 0)
            raw = t.cpu().tolist()
            if raw[0] == 0:
                raise RuntimeError("RCCL transport: rank 0 could not create the unique id" + (": " + err if err else ""))
            ida = C.create_string_buffer(bytes(raw[1:]), 128)
            rc = lib.GeneoRcclCreate(ida, int(plan.rank), int(plan.size), C.byref(self.h))
            msg = lib.GeneoRcclGetError().decode() if rc else ""
            ok = torch.tensor([0 if rc else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if not rc:
                    lib.GeneoRcclDestroy(C.byref(self.h))
                raise RuntimeError("RCCL transport: communicator creation failed on at least one rank" + (": " + msg if msg else ""))
            return
        if err:
            raise RuntimeError(err)
        if lib.GeneoRcclCreate(ida, int(plan.rank), int(plan.size), C.byref(self.h)):
            raise RuntimeError(lib.GeneoRcclGetError().decode())

    def attach(self, pc):
        import ctypes as C
        p = self.plan
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        owned, halo = i32(p.owned), i32(p.halo_gid)
        rc, sc, si = i32(p.recv_counts), i32(p.send_counts), i32(p.send_idx)
        ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        if self.lib.PCGenEOSetCommRccl(pc.h, self.h, len(owned), ptr(owned), len(halo), ptr(halo), ptr(rc), ptr(sc), ptr(si),
                                       self.WIDTH):
            raise RuntimeError(self.lib.GeneoRcclGetError().decode())
        pc.n_owned = len(owned)
        pc.rank = int(p.rank)

    def close(self):
        import ctypes as C
        if self.h:
            self.lib.GeneoRcclDestroy(C.byref(self.h))


def gather_owned(x_owned, plan, n_global):
    """Test helper: assemble a global numpy vector from every rank's owned part."""
    import torch
    import torch.distributed as dist
    out = [None] * plan.size
    dist.all_gather_object(out, (np.asarray(plan.owned), np.asarray(x_owned)))
    full = np.zeros(n_global)
    for own, val in out:
        full[own] = val
    return full
