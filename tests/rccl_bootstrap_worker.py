"""Worker of tests/test_gloo.py::test_rccl_bootstrap_failure_is_collective: two CPU ranks over gloo try to start the C++
RCCL transport of the test-only library (no GPU here: creating the unique id or the communicator fails).  What is
checked: every rank gets the RuntimeError together -- nobody is left waiting in a collective -- so that a caller can fall
back to another transport on all ranks (bench.py does)."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch.distributed as dist
    import hostsim_util as hu
    from geneo4petsc_amd import comm as gcomm
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    lib = hu.hostsim_lib()
    plan = types.SimpleNamespace(rank=rank, size=size)
    try:
        c = gcomm.RcclComm(plan, lib, dist, None)
        c.close()
        outcome = "created"         # a box with GPUs: fine as well, as long as every rank says the same
    except RuntimeError as e:
        outcome = "refused: %s" % str(e)[:60]
    seen = [None] * size
    dist.all_gather_object(seen, outcome.split(":")[0])
    assert len(set(seen)) == 1, seen
    dist.barrier()
    if rank == 0:
        print("OUTCOME " + outcome, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
