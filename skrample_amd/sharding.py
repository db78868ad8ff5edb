"""Batch sharding across GPUs: one process per GPU, contiguous batch slices, no data-path
collective.  Every reduction of the hot path is per sample, and the Philox key of a sample is its seed indexed by
GLOBAL sample id, so the concatenation of N shards equals the single-process result bit for bit.

This module is the one place that rule lives: bench.py, the tools and tests/test_sharding.py all import it.
It is plain Python (no HIP library needed) so the 2-rank gloo tests can drive it on CPU."""

from __future__ import annotations

import dataclasses
import os
import time
from collections.abc import Callable, Mapping


@dataclasses.dataclass(frozen=True)
class BatchShard:
    "the slice [first_sample, first_sample + batch) of a global batch of world * batch samples owned by `rank`"

    rank: int
    world: int
    batch: int  # samples per rank (weak scaling: fixed per-GPU work)
    local_rank: int = 0

    def __post_init__(self) -> None:
        if not (0 <= self.rank < self.world) or self.batch < 0:
            raise ValueError(f"bad shard: rank {self.rank} of {self.world}, batch {self.batch}")

    @classmethod
    def from_env(cls, batch: int, env: Mapping[str, str] | None = None) -> "BatchShard":
        "torch.distributed.run's environment (RANK / WORLD_SIZE / LOCAL_RANK); a plain run is rank 0 of 1"
        env = os.environ if env is None else env
        return cls(int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), batch, int(env.get("LOCAL_RANK", "0")))

    @property
    def first_sample(self) -> int:
        return self.rank * self.batch

    @property
    def global_batch(self) -> int:
        return self.world * self.batch

    @property
    def sample_ids(self) -> range:
        "global ids of this rank's samples"
        return range(self.first_sample, self.first_sample + self.batch)

    def seeds(self, base: int = 42) -> list[int]:
        "per-sample noise seeds: a function of the GLOBAL sample id only, so results do not depend on the shard count"
        return [base + i for i in self.sample_ids]

    def input_seed(self, base: int = 1234) -> int:
        "seed of this rank's synthetic-input generator (distinct data per shard)"
        return base + self.first_sample


def max_over_ranks(values: list[float], dist=None, device=None) -> list[float]:
    """The benchmark's only cross-rank exchange: element-wise MAX of a few timings (the slowest rank defines the step
    time).  `dist` is torch.distributed (None or uninitialised = single process)."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return list(values)
    import torch

    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


def _group_ready(dist) -> bool:
    return dist is not None and dist.is_available() and dist.is_initialized()


class TimedRegion:
    """The benchmark's timed window on one rank.  `open()` aligns the ranks (barrier + device synchronise) and starts the
    clock; `close()` synchronises THIS rank's device, reads the clock, and only then joins the closing barrier -- so the
    wall time a rank reports holds its own work and nothing of the collective ("a single barrier ... for
    timing alignment only"; the engine itself has no exchange step to bill).  The slowest rank still defines the job's
    step time: that is `max_over_ranks` over the walls, taken after the region.

    `sync` = the rank's device synchronise (torch.cuda.synchronize on a GPU rank, nothing on a CPU rehearsal),
    `barrier` = the collective (default dist.barrier; tests pass one that dawdles)."""

    def __init__(self, dist=None, sync: Callable[[], None] | None = None, barrier: Callable[[], None] | None = None, clock: Callable[[], float] = time.perf_counter):
        self.sync = sync or (lambda: None)
        self.barrier = barrier or (dist.barrier if _group_ready(dist) else (lambda: None))
        self.clock = clock
        self.t0: float | None = None
        self.wall: float | None = None

    def open(self) -> float:
        self.barrier()
        self.sync()
        self.t0 = self.clock()
        return self.t0

    def close(self) -> float:
        assert self.t0 is not None, "close() before open()"
        self.sync()
        self.wall = self.clock() - self.t0  # read BEFORE the closing collective
        self.barrier()
        self.sync()
        return self.wall


def rank_spread(values: Mapping[str, float], dist=None, device=None) -> dict:
    """Per-rank view of a few timings for the bench line (`roofline.ranks`): for each key the min, the max and every rank's
    own value in rank order, plus the number of ranks the process group actually holds -- skew between ranks is then visible
    in the one JSON line instead of hidden inside a MAX."""
    keys = list(values)
    mine = [float(values[k]) for k in keys]
    if not _group_ready(dist):
        rows, seen = [mine], 1
    else:
        import torch

        t = torch.tensor(mine, dtype=torch.float64, device=device)
        got = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(got, t)
        rows, seen = [g.tolist() for g in got], dist.get_world_size()
    out: dict = {"n_ranks_seen": seen}
    for j, k in enumerate(keys):
        col = [r[j] for r in rows]
        out[k] = {"min": min(col), "max": max(col), "per_rank": col}
    return out


def aggregate_rate(units_per_rank: int, world: int, seconds: float) -> float:
    "whole-job throughput: the units all ranks processed / the slowest rank's time"
    return world * units_per_rank / seconds
