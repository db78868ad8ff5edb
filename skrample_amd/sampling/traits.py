"""Configuration mix-ins that make up a sampler's public, frozen-dataclass config surface.

These are the knobs of reference `skrample/sampling/traits.py:9-61` under the same names, so that
`DPM(order=2, stochasticity=1, derivative_transform=...)` constructs identically:

  order                 (HigherOrder)          requested solver order; the order actually used ramps up over
                                               the first steps and back down at the end of the schedule
  stochasticity         (Stochastic)           eta: 0 = deterministic ODE step, 1 = full SDE step
  derivative_transform  (DerivativeTransform)  prediction space the multistep / Runge-Kutta arithmetic runs in
                                               (default x-hat-0; None = the model's own space)

They carry no tensor code: on this engine a sampler turns these numbers into kernel coefficients.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass

from ..common import Point
from .models import DataModel, DiffusionModel

_X_HAT_SPACE = DataModel()  # shared default instance: conversions short-circuit on identity (models.ModelConvert)


@dataclass(frozen=True)
class Stochastic:
    stochasticity: float = 0


@dataclass(frozen=True)
class DerivativeTransform:
    derivative_transform: DiffusionModel | None = _X_HAT_SPACE


@dataclass(frozen=True)
class HigherOrder(ABC):
    order: int = 2

    @staticmethod
    @abstractmethod
    def max_order() -> int:
        "largest order the solver implements"

    @staticmethod
    def min_order() -> int:
        "smallest order the solver will use"
        return 1


@dataclass(frozen=True)
class UnifiedModelling(DerivativeTransform, Stochastic, HigherOrder):
    "order + stochasticity + derivative space in one MRO-stable bundle (field order: order, stochasticity, derivative_transform)"


@dataclass(frozen=True)
class SamplingCommon:
    "noising helpers every sampler exposes; both are plain forwards to `Point`"

    def remove_noise(self, sample, noise, point: Point):
        return point.remove_noise(sample, noise)

    def add_noise(self, sample, noise, point: Point):
        return point.add_noise(sample, noise)
