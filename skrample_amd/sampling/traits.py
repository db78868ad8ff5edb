"""Configuration mix-ins shared by the samplers (reference `skrample/sampling/traits.py:9-61`):
they define the public, frozen-dataclass config surface (order, stochasticity, derivative space)."""

from __future__ import annotations

import abc
import dataclasses

from .. import common
from . import models


@dataclasses.dataclass(frozen=True)
class SamplingCommon:
    def add_noise(self, sample, noise, point: common.Point):
        return point.add_noise(sample, noise)

    def remove_noise(self, sample, noise, point: common.Point):
        return point.remove_noise(sample, noise)


@dataclasses.dataclass(frozen=True)
class HigherOrder(abc.ABC):
    order: int = 2
    "requested solver order; the order actually used ramps up at the start and down at the end"

    @staticmethod
    def min_order() -> int:
        return 1

    @staticmethod
    @abc.abstractmethod
    def max_order() -> int: ...


@dataclasses.dataclass(frozen=True)
class Stochastic:
    stochasticity: float = 0
    "eta: 0 = deterministic ODE, 1 = full SDE"


@dataclasses.dataclass(frozen=True)
class DerivativeTransform:
    derivative_transform: models.DiffusionModel | None = models.DataModel()  # noqa: RUF009 - immutable
    "space in which the multistep / Runge-Kutta arithmetic happens (None = the model's own space)"


@dataclasses.dataclass(frozen=True)
class UnifiedModelling(DerivativeTransform, Stochastic, HigherOrder):
    "order + stochasticity + derivative space, in one MRO-stable bundle"
