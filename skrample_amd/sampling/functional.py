"""Closure-driven (functional) samplers: the sampler owns the loop and calls the model itself.

Public surface follows reference `skrample/sampling/functional.py`: step_tableau (:55-105),
FunctionalSampler.sample_model / generate_model (:108-149), RKUltra (:217-268), DynasauRK (:271-349)
RKMoire (:352-472, adaptive: one fused launch for the embedded pair, one reduction launch + read-back per error
norm) and the provider maps (:18-52).

One Runge-Kutta step with s stages costs exactly s fused launches: every stage input
X_j = Gamma*x0 + Delta * (sum_i a_ji d_i)/sum_i a_ji and the final combination are single lazy
forms over the *aliased* pairs (X_i, model(X_i)) -- the derivative conversions d_i = ws*X_i + wo*out_i
are folded into the coefficients instead of being materialised.
"""

from __future__ import annotations

import dataclasses
import math
from abc import ABC, abstractmethod
from types import MappingProxyType
from typing import Any, Callable, Mapping

from .. import common, scheduling
from ..common import DeltaPoint, Point, Step
from . import lazy, models, tableaux, traits
from .lazy import Lin, lift

SampleCallback = Callable[[Any, int, DeltaPoint], Any]
SampleableModel = Callable[[Any, float, float, float], Any]

DEFAULT_PROVIDERS: Mapping[int, Any] = {
    1: tableaux.RK1.Euler,
    2: tableaux.RK2.Mid,
    3: tableaux.RK2.EES5_MIN,
    4: tableaux.RK2.EES7_MIN,
    5: tableaux.SSP.RK4_5,
    6: tableaux.RKE5.CashKarp,
    7: tableaux.RKZ.Butcher6,
    8: tableaux.SSP.RK3_8,
    10: tableaux.SSP.RK5_10,
    11: tableaux.RKZ.CV8,
    15: tableaux.RKZ.Stepanov10,
}
"default tableau per *stage count* (not mathematical order)"

STABLE_PROVIDERS: Mapping[int, Any] = {
    2: tableaux.RKE2.Heun,
    3: tableaux.SSP.RK3_3,
    4: tableaux.RKE3.SSPRK3_4,
    5: tableaux.SSP.RK3_5,
    6: tableaux.SSP.RK3_6,
    7: tableaux.SSP.RK3_7,
}

DEFAULT_EMBEDDED_PROVIDERS: Mapping[int, Any] = {
    2: tableaux.RKE2.Heun,
    4: tableaux.RKE3.BogackiShampine,
    6: tableaux.RKE5.Fehlberg,
}


def _materialise(value, like):
    return lazy.settle(value, like=like) if isinstance(value, Lin) else value


def step_tableau(
    tableau,
    sample,
    model: SampleableModel,
    model_transform: models.DiffusionModel,
    schedule: scheduling.SkrampleSchedule,
    step: Step,
    derivative_transform: models.DiffusionModel | None = None,
    noise=None,
    stochasticity: float = 0,
    epsilon: float = 1e-8,
) -> tuple:
    """One explicit Runge-Kutta step in derivative space; returns one result per weight row."""
    from . import native

    done = native.step_tableau(tableau, sample, model, model_transform, schedule, step, derivative_transform, noise, stochasticity, epsilon)
    if done is not None:  # a 16-bit tensor sample outside a compute scale: the reference's rounded tensor ops
        return done
    nodes, weight_rows = tableau[0], tableau[1:]
    convert = models.ModelConvert(model_transform, derivative_transform) if derivative_transform else None
    space = derivative_transform or model_transform

    t0, t1 = step
    p0, p1, *stage_points = schedule.ipoints([t0, t1, *(t0 + c * (t1 - t0) for c, _ in nodes)])
    delta = DeltaPoint(p0, p1)
    base = lift(sample)
    derivs: list = []
    for point, (_, row) in zip(stage_points, nodes):
        if row:
            mix = common.sumprod(derivs, row) / math.fsum(row)
            stage_in = _materialise(space.update_form(base, mix, DeltaPoint(p0, point)), sample)
        else:
            stage_in = sample
        if abs(point.timestep) < epsilon or abs(point.sigma) < epsilon:
            # never evaluate the network at the clean end: use the derivative that would reproduce X
            g, d = space.gamma(delta), space.delta(delta)
            derivs.append((lift(stage_in) - base * g) / d)
        else:
            out = model(stage_in, *point)
            derivs.append(convert.form_to(stage_in, out, point) if convert else lift(out))
    forms = [space.update_form(base, common.sumprod(derivs, row), delta, noise, stochasticity) for row in weight_rows]
    if len(forms) == 2 and all(isinstance(f, Lin) for f in forms):  # embedded pair: both solutions from ONE launch
        dtype = sample.dtype if isinstance(sample, lazy.torch.Tensor) else None  # (ndarrays: the forms' own leaves name the dtype)
        return tuple(lazy.evaluate(forms, [dtype, dtype]))
    return tuple(_materialise(f, sample) for f in forms)


@dataclasses.dataclass(frozen=True)
class FunctionalSampler(ABC, traits.SamplingCommon):
    @abstractmethod
    def sample_model(self, sample, model, model_transform, schedule, steps: int, include: slice = slice(None), rng=None, callback=None):
        "run the steps selected by `include` (of `steps` total) on an already-noised sample"

    def generate_model(self, model, model_transform, schedule, rng, steps: int, include: slice = slice(None), initial=None, callback=None):
        "like sample_model, but draws (and scales) the starting noise itself"
        if initial is None and include.start is None:
            sample = rng(None)
        else:
            start = schedule.ipoint((include.start or 0) / steps)
            noise = rng(None)
            mixed = lift(noise) * start.sigma if initial is None else lift(initial) * start.alpha + lift(noise) * start.sigma
            full = schedule.point_1
            sample = lazy.settle(mixed / (0.0 * full.alpha + 1.0 * full.sigma), like=noise)
        return self.sample_model(sample, model, model_transform, schedule, steps, include, rng, callback)


@dataclasses.dataclass(frozen=True)
class FunctionalHigher(traits.HigherOrder, FunctionalSampler):
    def adjust_steps(self, steps: int) -> int:
        return round(steps / self.order)


@dataclasses.dataclass(frozen=True)
class FunctionalUnified(traits.UnifiedModelling, FunctionalHigher): ...


@dataclasses.dataclass(frozen=True)
class FunctionalSinglestep(FunctionalSampler):
    @abstractmethod
    def step(self, sample, model, model_transform, schedule, step: Step, rng=None): ...

    def sample_model(self, sample, model, model_transform, schedule, steps, include=slice(None), rng=None, callback=None):
        for n in list(range(steps))[include]:
            step = Step.from_int(n, steps)
            sample = self.step(sample, model, model_transform, schedule, step, rng)
            if callback:
                callback(sample, n, schedule.istep(step))
        return sample


def _error_mean(a, b, power: int) -> float:
    "mean(|a - b|^power): python numbers on the host, device tensors through skr_error_mean"
    if isinstance(b, (int, float)) and isinstance(a, (int, float)):
        return abs(a - b) ** power
    return lazy.error_mean(a, b, power)


@dataclasses.dataclass(frozen=True)
class FunctionalAdaptive(FunctionalSampler):
    "samplers that choose their own step size from an error estimate"

    Evaluator = Callable[[Any, Any], float]
    "signature of an error measure between two samples"

    @staticmethod
    def mae(a, b) -> float:
        return _error_mean(a, b, 1)

    @staticmethod
    def mse(a, b) -> float:
        return _error_mean(a, b, 2)

    evaluator: Callable[[Any, Any], float] = mse
    threshold: float = 1e-2


@dataclasses.dataclass(frozen=True)
class RKUltra(FunctionalUnified, FunctionalSinglestep):
    "fixed-tableau explicit Runge-Kutta; `order` selects the tableau with the most stages <= order"

    providers: Mapping[int, Any] = MappingProxyType(DEFAULT_PROVIDERS)

    @staticmethod
    def max_order() -> int:
        return 99

    def tableau(self, order: int | None = None) -> tableaux.Tableau:
        order = self.order if order is None else order
        usable = [k for k in self.providers if k <= order]
        if order >= min(self.providers) and usable and max(usable):
            tab = self.providers[max(usable)].tableau()
            return tableaux.Tableau(tab.stages, tab.weights)
        return tableaux.RK1.Euler.value

    def adjust_steps(self, steps: int) -> int:
        "steps that spend about the same number of model calls (stages at c = 1 are free on the last step)"
        stages = self.tableau()[0]
        return max(round(steps / len(stages) + sum(abs(1 - c) < 1e-8 for c, _ in stages) / len(stages)), 1)

    def step(self, sample, model, model_transform, schedule, step, rng=None):
        return step_tableau(self.tableau(), sample, model, model_transform, schedule, step, self.derivative_transform, rng(step) if rng else None, self.stochasticity)[0]


@dataclasses.dataclass(frozen=True)
class DynasauRK(FunctionalUnified, FunctionalSinglestep):
    """Runge-Kutta with a tableau generated per step: starts at the most stable member of a
    one-parameter family and decays exponentially towards the most convergent one
    (gradient = exp(-(S*T + s*t) * stages), T = total steps, t = current step)."""

    per_step_decay: float = math.log(0.5) / -2
    total_step_decay: float = math.log(0.5) / -20
    invert: bool = False

    @staticmethod
    def min_order() -> int:
        return 2

    @staticmethod
    def max_order() -> int:
        return 4

    def adjust_steps(self, steps: int) -> int:
        return max(round(steps / self.order), 1)

    def _family(self):
        if self.order >= 4:
            return 1 / 4 * (2 - math.sqrt(2)), 1 / 14 * (5 - 3 * math.sqrt(2)), tableaux.ees27_tableau
        if self.order >= 3:
            return 0.25, 0.1, tableaux.ees25_tableau
        return 1, 0.5, tableaux.rk2_tableau

    def gradient(self, step: Step, stages: int) -> float:
        step = step.normal().clamp()
        g = math.exp((-self.total_step_decay * step.amount() - self.per_step_decay * step.position()) * stages)
        return abs(self.invert - min(max(g, 0), 1))

    def tableau(self, step: Step) -> tableaux.Tableau:
        stable, convergent, make = self._family()
        g = self.gradient(step, len(make((stable + convergent) / 2).stages))
        return make(g * stable + (1 - g) * convergent)

    def step(self, sample, model, model_transform, schedule, step, rng=None):
        return step_tableau(self.tableau(step), sample, model, model_transform, schedule, step, self.derivative_transform, rng(step) if rng else None, self.stochasticity)[0]


@dataclasses.dataclass(frozen=True)
class RKMoire(traits.DerivativeTransform, FunctionalAdaptive, FunctionalHigher):
    """Adaptive Runge-Kutta over an embedded pair: every step yields a high- and a low-order solution (one fused
    launch for both), their relative error (one reduction launch each, read back) sets the next step size."""

    providers: Mapping[int, Any] = MappingProxyType(DEFAULT_EMBEDDED_PROVIDERS)
    threshold: float = 1e-4
    initial: float = 1 / 50
    "first step, as a fraction of the schedule"
    maximum: float = 1 / 4
    "largest step, as a fraction of the schedule"
    adaption: float = 0.3
    "exponent of the step-size controller"
    discard: float = float("inf")
    "redo a step whose size has to shrink by more than this factor"
    rescale_init: bool = True
    rescale_max: bool = False

    @staticmethod
    def min_order() -> int:
        return 2

    @staticmethod
    def max_order() -> int:
        return 99

    def adjust_steps(self, steps: int) -> int:
        return steps

    def tableau(self, order: int | None = None) -> tableaux.EmbeddedTableau:
        order = self.order if order is None else order
        usable = [k for k in self.providers if k <= order]
        if order >= min(self.providers) and usable and max(usable):
            return self.providers[max(usable)].tableau()
        return tableaux.RKE2.Heun.tableau()

    def sample_model(self, sample, model, model_transform, schedule, steps, include=slice(None), rng=None, callback=None):
        pair = self.tableau()
        first, largest = self.initial, self.maximum
        if self.rescale_init:
            first *= len(pair.stages) / 2  # relative to Heun's two evaluations
        if self.rescale_max:
            largest *= len(pair.stages) / 2
        stride = max(round(steps * first), 1)
        tiny = 1e-16
        wanted = list(range(steps))[include]
        at = wanted[0]
        while at <= wanted[-1]:
            upto = min(at + stride, wanted[-1] + 1)
            if upto < steps:
                high, low = step_tableau(pair, sample, model, model_transform, schedule, Step(at / steps, upto / steps), self.derivative_transform)
                s0, s1, s2 = schedule.ipoints_np([at / steps, upto / steps, (upto + stride) / steps])[:, 1].tolist()
                slope = abs(s0 - s1) / abs(s1 - s2)  # how much the next sigma step grows by itself
                error = self.evaluator(low, high) / max(self.evaluator(0, high), tiny)
                adjustment = (self.threshold / max(error, tiny)) ** self.adaption / slope
                stride = max(round(min(stride * adjustment, steps * largest)), 1)
                if upto - at > stride and 1 / max(adjustment, tiny) > self.discard:
                    continue  # too optimistic: retry from the same point with the smaller stride
            else:  # last stretch: the error estimate would not be used
                high = step_tableau(pair.unembed(), sample, model, model_transform, schedule, Step(at / steps, 1), self.derivative_transform)[0]
            sample = high
            if callback:
                callback(sample, upto - 1, schedule.istep(Step.from_int(at, steps)))
            at = upto
        return sample
