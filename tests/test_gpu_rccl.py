"""-m gpu: the C++ RCCL transport of libgeneopc (csrc/comm_rccl.cpp) on the one GPU of the test box.

RCCL refuses two ranks on one device, so what can run here is a ONE-rank communicator: unique id, ncclCommInitRank,
the halo exchange as an ncclSend / ncclRecv group (to itself, with the real offset arithmetic: per-peer offsets, widths,
forward and reverse roles) and ncclAllReduce, all on the library stream, followed by a full GenEO solve with the
transport attached.  The N > 1 data movement has the same code path with more peers; its semantics are pinned on CPU by
the gloo tests (tests/test_gloo.py) through comm.TorchComm, which this transport mirrors line by line."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


class Plan:
    rank, size = 0, 1

    def __init__(self, n, k):
        self.owned = np.arange(n, dtype=np.int32)
        self.halo_gid = np.zeros(0, dtype=np.int32)
        self.recv_counts = np.array([k], dtype=np.int32)
        self.send_counts = np.array([k], dtype=np.int32)
        self.send_idx = np.arange(k, dtype=np.int32)


@pytest.fixture(scope="module")
def lib():
    from geneo4petsc_amd import _lib
    return _lib.load()


def test_rccl_transport_one_rank(lib):
    from geneo4petsc_amd.comm import RcclComm
    from geneo4petsc_amd.pc import GenEOPC
    k = 37
    comm = RcclComm(Plan(100, k), lib)
    pc = GenEOPC(lib)
    pc.set_sizes(100, 1)
    comm.attach(pc)
    send, recv, red = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.GeneoRcclPlanBuffers(comm.h, 0, C.byref(send), C.byref(recv), C.byref(red)) == 0
    for flag, w in ((0, 1), (1, 1), (0 | (32 << 1), 32), (1 | (5 << 1), 5)):
        src = np.random.default_rng(flag).random(k * w)
        assert lib.GeneoH2D(send, src.ctypes.data_as(C.c_void_p), src.nbytes) == 0
        assert lib.GeneoRcclPlanExchange(comm.h, 0, flag) == 0, lib.GeneoRcclGetError().decode()
        lib.GeneoDeviceSync()
        got = np.zeros(k * w)
        assert lib.GeneoD2H(got.ctypes.data_as(C.c_void_p), recv, got.nbytes) == 0
        np.testing.assert_array_equal(got, src)
    vals = np.random.default_rng(9).random(1000)
    lib.GeneoH2D(red, vals.ctypes.data_as(C.c_void_p), vals.nbytes)
    assert lib.GeneoRcclPlanAllreduce(comm.h, 0, 1000) == 0, lib.GeneoRcclGetError().decode()
    lib.GeneoDeviceSync()
    got = np.zeros(1000)
    lib.GeneoD2H(got.ctypes.data_as(C.c_void_p), red, got.nbytes)
    np.testing.assert_array_equal(got, vals)
    assert lib.GeneoRcclPlanExchange(comm.h, 0, 0 | (64 << 1)) != 0        # wider than the buffers: refused, not overrun
    pc.destroy()
    comm.close()
