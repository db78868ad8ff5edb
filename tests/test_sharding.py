"""N > 1 path of bench.py on CPU: two gloo ranks agree on the shard arithmetic (contiguous batch slices,
seeds by GLOBAL sample index, max-over-ranks timing) -- the only things ranks share; there is no data-path
collective to test."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skr_oracle import noise as ON

B_PER_RANK, UNIT = 4, 64


def shard_seeds(rank: int, batch: int) -> list[int]:
    "same rule as bench.py::capture_plans"
    return [42 + rank * batch + i for i in range(batch)]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = shard_seeds(rank, B_PER_RANK)
    # each rank "generates" its shard's step noise from the RNG specification (what the kernel draws)
    shard = np.stack([ON.philox_normal(s, 3 * 256, UNIT) for s in seeds])
    elapsed = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), shard)
    if rank == 0:
        np.save(os.path.join(out_dir, "elapsed.npy"), elapsed.numpy())
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole = np.stack([ON.philox_normal(s, 3 * 256, UNIT) for s in shard_seeds(0, 2 * B_PER_RANK)])
    parts = np.concatenate([np.load(tmp_path / "shard0.npy"), np.load(tmp_path / "shard1.npy")])
    assert np.array_equal(whole, parts)  # 1 process over the full batch == concatenation of the 2 shards
    assert np.load(tmp_path / "elapsed.npy")[0] == 0.020  # rank 0 reports the slowest rank's time
