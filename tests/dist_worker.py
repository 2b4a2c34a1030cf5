"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-2 gloo job.

Mirrors the multi-GPU path of bench.py on CPU: shard the batch, advance the local shard (with
the CPU emulation of the kernel core -- test infrastructure), gather the final state through
the launcher's process group, and check it on rank 0 against the unsharded result."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch.distributed as dist  # noqa: E402

import emul_binding as eb  # noqa: E402
import helpers  # noqa: E402
from jaxsim_amd import distributed  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    zoo = helpers.ModelZoo()
    model = zoo("anymal")
    N = 2 * 6
    d = zoo.random_data("anymal", N, seed=3)
    full = helpers.odata_to_block(model, d)
    lo, hi = distributed.shard_bounds(N, rank, world)
    local = distributed.shard_block(full, rank, world)
    assert local.shape[1] == hi - lo
    for _ in range(3):
        local = eb.run(model, eb.MODE_STEP, local)
    gathered = distributed.all_gather_state_blocks_host(local)
    # the torch-free bootstrap of bench.py: every rank of this launch derives the same job key, and the
    # host-side file collective (the fallback when RCCL is unavailable) gathers one scalar per rank
    keys = [None] * world
    dist.all_gather_object(keys, distributed.job_key())
    assert len(set(keys)) == 1 and keys[0].endswith(str(os.getppid())), keys
    fc = distributed.FileCollective(rank, world, "test_" + keys[0])
    got = fc.all_gather_scalars(10.0 + rank)
    assert got.tolist() == [10.0 + r for r in range(world)], got
    if rank == 0:
        ref = full
        for _ in range(3):
            ref = eb.run(model, eb.MODE_STEP, ref)
        assert gathered.shape == ref.shape
        np.testing.assert_array_equal(gathered, ref)
        print("DIST_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
