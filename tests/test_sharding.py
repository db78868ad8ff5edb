"""N > 1 path on CPU: two and four gloo ranks run the SAME shard arithmetic bench.py runs (skrample_amd.sharding: rank / world
from the launcher's environment, contiguous batch slices, seeds by GLOBAL sample index, MAX-over-ranks timing, whole-job
rate) -- the only things ranks share; there is no data-path collective to test."""

import ast
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skr_oracle import noise as ON
from skrample_amd.sharding import BatchShard, aggregate_rate, max_over_ranks

B_PER_RANK, UNIT = 4, 64


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    # what torch.distributed.run exports for bench.py
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = BatchShard.from_env(B_PER_RANK)
    assert (shard.rank, shard.world, shard.local_rank) == (rank, world, rank)
    # each rank "generates" its shard's step noise from the RNG specification (what the kernel draws)
    noise = np.stack([ON.philox_normal(s, 3 * 256, UNIT) for s in shard.seeds()])
    dist.barrier()
    wall, kernel = max_over_ranks([0.010 * (rank + 1), 0.002 * (world - rank)], dist)
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), noise)
    np.save(os.path.join(out_dir, f"ids{rank}.npy"), np.array(list(shard.sample_ids)))
    if rank == 0:
        np.save(os.path.join(out_dir, "summary.npy"), np.array([wall, kernel, aggregate_rate(20, world, wall), shard.global_batch]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_rank_shards_equal_single_process(world, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    single = BatchShard(0, 1, world * B_PER_RANK)
    whole = np.stack([ON.philox_normal(s, 3 * 256, UNIT) for s in single.seeds()])
    parts = np.concatenate([np.load(tmp_path / f"shard{r}.npy") for r in range(world)])
    assert np.array_equal(whole, parts)  # 1 process over the full batch == concatenation of the shards
    assert np.concatenate([np.load(tmp_path / f"ids{r}.npy") for r in range(world)]).tolist() == list(single.sample_ids)
    wall, kernel, rate, global_batch = np.load(tmp_path / "summary.npy")
    assert wall == pytest.approx(0.010 * world) and kernel == pytest.approx(0.002 * world)  # rank 0 reports the slowest rank's times
    assert rate == pytest.approx(world * 20 / (0.010 * world)) and global_batch == world * B_PER_RANK


def test_shard_rules():
    s = BatchShard.from_env(256, {"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"})
    assert (s.first_sample, s.global_batch, s.seeds()[0], s.seeds()[-1], s.input_seed()) == (768, 2048, 42 + 768, 42 + 1023, 1234 + 768)
    assert BatchShard.from_env(256, {}) == BatchShard(0, 1, 256, 0)
    assert max_over_ranks([1.5, 2.5]) == [1.5, 2.5]  # single process: no process group
    for bad in ((2, 2, 4), (-1, 2, 4), (0, 1, -1)):
        try:
            BatchShard(*bad)
        except ValueError:
            continue
        raise AssertionError(bad)


def test_bench_uses_the_shared_rule():
    "bench.py takes rank / world / seeds / timing reduction from skrample_amd.sharding and nowhere else"
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)
    imported = {a.name for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module == "skrample_amd.sharding" for a in n.names}
    assert {"BatchShard", "aggregate_rate", "max_over_ranks"} <= imported
    assert 'os.environ.get("RANK"' not in src and "all_reduce" not in src and "rank * batch" not in src
