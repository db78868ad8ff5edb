"""Batch sharding across GPUs: one process per GPU, contiguous batch slices, no data-path
collective.  Every reduction of the hot path is per sample, and the Philox key of a sample is its seed indexed by
GLOBAL sample id, so the concatenation of N shards equals the single-process result bit for bit.

This module is the one place that rule lives: bench.py, the tools and tests/test_sharding.py all import it.
It is plain Python (no HIP library needed) so the 2-rank gloo tests can drive it on CPU."""

from __future__ import annotations

import dataclasses
import os
from collections.abc import Mapping


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


def aggregate_rate(units_per_rank: int, world: int, seconds: float) -> float:
    "whole-job throughput: the units all ranks processed / the slowest rank's time"
    return world * units_per_rank / seconds
