"""N > 1 path on CPU: two, four and eight gloo ranks (eight = the node the driver measures) run the SAME shard arithmetic bench.py runs (skrample_amd.sharding: rank / world
from the launcher's environment, contiguous batch slices, seeds by GLOBAL sample index, MAX-over-ranks timing, whole-job
rate) -- the only things ranks share; there is no data-path collective to test."""

import ast
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skr_oracle import noise as ON
from skrample_amd.sharding import BatchShard, TimedRegion, aggregate_rate, max_over_ranks, rank_spread

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


@pytest.mark.parametrize("world", [2, 4, 8])
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


def _timing_worker(rank: int, world: int, port: int, out_dir: str, dawdle_s: float) -> None:
    """bench.py's timed window with synthetic "work" (a fixed sleep per rank): the LAST rank dawdles for `dawdle_s` inside
    its closing barrier (it enters the collective late), which holds every other rank inside theirs for that long"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = [0]

    def barrier() -> None:
        calls[0] += 1
        if calls[0] == 2 and rank == world - 1:  # the closing one
            time.sleep(dawdle_s)
        dist.barrier()

    steps, work_s = 20, 0.020 + 0.002 * rank
    region = TimedRegion(dist, barrier=barrier)
    region.open()
    time.sleep(work_s)
    left = time.perf_counter()
    wall = region.close()
    held = time.perf_counter() - left  # how long close() kept this rank (the collective)
    spread = rank_spread({"wall_us_per_step": wall * 1e6 / steps}, dist)
    (slowest,) = max_over_ranks([wall], dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "timing.npy"), np.array([wall, slowest, held, aggregate_rate(steps, world, slowest), spread["n_ranks_seen"], *spread["wall_us_per_step"]["per_rank"]]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_closing_barrier_is_not_billed_to_the_timed_steps(world, tmp_path):
    """A rank that sits 50 ms in the closing collective changes nobody's wall time and so not `value` (the
    old window read the clock after dist.barrier(), so an 8-rank RCCL barrier was billed to a 0.54 ms K = 20 region)."""
    rates = {}
    for dawdle in (0.0, 0.050):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = tmp_path / f"d{int(dawdle * 1e3)}"
        out.mkdir()
        mp.spawn(_timing_worker, args=(world, port, str(out), dawdle), nprocs=world, join=True)
        wall0, slowest, held0, rate, seen, *per_rank = np.load(out / "timing.npy")
        assert seen == world and len(per_rank) == world
        slowest_work = 0.020 + 0.002 * (world - 1)
        assert slowest_work <= slowest < slowest_work + 0.012  # the slowest rank's own work, nothing of the barrier
        assert 0.020 <= wall0 < 0.032 and max(per_rank) == pytest.approx(slowest * 1e6 / 20)
        if dawdle:
            assert held0 >= 0.045  # rank 0 really did sit in the collective for the dawdler's 50 ms ...
        rates[dawdle] = rate
    assert rates[0.050] == pytest.approx(rates[0.0], rel=0.25) and rates[0.050] > world * 20 / 0.040  # ... and `value` did not move (it would halve)


def test_timed_region_single_process():
    ticks = iter([10.0, 10.5])
    log = []
    r = TimedRegion(None, sync=lambda: log.append("sync"), clock=lambda: next(ticks))
    r.open()
    assert r.close() == 0.5 and log == ["sync"] * 3  # open: sync; close: sync before the clock read, sync after the (absent) barrier
    assert rank_spread({"a": 2.0}) == {"n_ranks_seen": 1, "a": {"min": 2.0, "max": 2.0, "per_rank": [2.0]}}


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
    assert {"BatchShard", "TimedRegion", "aggregate_rate", "max_over_ranks", "rank_spread"} <= imported
    assert "dist.barrier()\n    wall" not in src and "wall = region.close()" in src  # the clock is read inside TimedRegion.close, before the collective
    assert 'os.environ.get("RANK"' not in src and "all_reduce" not in src and "rank * batch" not in src
