"""Noise schedules: continuous maps t in [0,1] -> (timestep, sigma, alpha), fp64 numpy on the host.

Per BASELINE.json `north_star` the sigma tables stay host-side; this module keeps the class names,
constructor fields and methods of reference `skrample/scheduling.py` (SigmaSpace :22-48,
SkrampleSchedule :65-135, ScheduleCommon :138-157, FixedSchedule :160-177, Scaled :180-251,
ZSNR :254-278, Linear :281-317, SubSchedule/SubSigmas :342-380, ScheduleModifier :383-474,
Karras/Exponential/Beta/Probit :493-580, FlowShift/Hyper/Sinner :583-664) so schedules built for
skrample construct unchanged.  All schedule objects are frozen, hashable dataclasses (they key the
LRU caches below and the wrapper's per-step coefficient cache).
"""

from __future__ import annotations

import dataclasses
import functools
import math
from abc import ABC, abstractmethod
from dataclasses import dataclass, replace
from typing import Sequence

import numpy as np

from .common import DeltaPoint, Point, Step, normalize, regularize, rescale_positive, sigmoid  # (normalize / regularize / sigmoid are re-exported, as in the reference)

NPPoints = np.ndarray
NPSequence = np.ndarray


# ---------------------------------------------------------------------------------------------------
# sigma spaces
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class SigmaSpace(ABC):
    @abstractmethod
    def normalize(self, regular_sigmas):
        "regular sigma -> (sigma, alpha) with the space's normalisation"

    @abstractmethod
    def regularize(self, normal_sigmas):
        "inverse of normalize on the sigma component"


@dataclass(frozen=True)
class VariancePreserving(SigmaSpace):
    "sigma^2 + alpha^2 = 1  (polar angle of the regular sigma)"

    def normalize(self, regular_sigmas):
        angle = np.arctan(regular_sigmas)
        return np.sin(angle), np.cos(angle)

    def regularize(self, normal_sigmas):
        return np.tan(np.arcsin(normal_sigmas))


@dataclass(frozen=True)
class FlowMatching(SigmaSpace):
    "sigma + alpha = 1"

    def normalize(self, regular_sigmas):
        sig = np.asarray(regular_sigmas)
        return sig, 1 - sig

    def regularize(self, normal_sigmas):
        return np.asarray(normal_sigmas)


# ---------------------------------------------------------------------------------------------------
# caches
# ---------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def np_schedule_lru(schedule: "SkrampleSchedule", steps: int) -> NPPoints:
    return schedule.schedule_np(steps)


@functools.lru_cache(maxsize=None)
def schedule_lru(schedule: "SkrampleSchedule", steps: int) -> Sequence[Point]:
    return tuple(Point(*row) for row in np_schedule_lru(schedule, steps).tolist())


@functools.lru_cache(maxsize=65536)
def ipoint_lru(schedule: "SkrampleSchedule", t: float) -> Point:
    "cached `schedule.ipoint(t)`: the samplers ask for the same handful of points every step"
    return schedule.ipoint(t)


def _unit(t) -> np.ndarray:
    return np.asarray(t, dtype=np.float64).clip(0, 1)


def _rows(arr: np.ndarray) -> list[Point]:
    return [Point(*row) for row in arr.tolist()]


# ---------------------------------------------------------------------------------------------------
# protocol
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class SkrampleSchedule(ABC):
    @property
    @abstractmethod
    def space(self) -> SigmaSpace: ...

    @abstractmethod
    def _points(self, t: NPSequence) -> NPPoints:
        "rows (timestep, sigma, alpha) at noise levels t (1 = all noise)"

    # noise-time view (t=1 all noise)
    def points_np(self, t) -> NPPoints:
        return self._points(_unit(t))

    def points(self, t) -> Sequence[Point]:
        return _rows(self.points_np(t))

    def point(self, t: float) -> Point:
        return Point(*self._points(np.expand_dims(np.float64(t).clip(0, 1), 0))[0].tolist())

    # inference-time view (t=0 all noise)
    def ipoints_np(self, t) -> NPPoints:
        return self._points(1 - _unit(t))

    def ipoints(self, t) -> Sequence[Point]:
        return _rows(self.ipoints_np(t))

    def ipoint(self, t: float) -> Point:
        return Point(*self._points(np.expand_dims(1 - np.float64(t).clip(0, 1), 0))[0].tolist())

    @functools.cached_property
    def point_0(self) -> Point:
        return self.point(0)

    @functools.cached_property
    def point_1(self) -> Point:
        return self.point(1)

    def step(self, step: Step) -> DeltaPoint:
        return DeltaPoint(*self.points(step))

    def istep(self, step: Step) -> DeltaPoint:
        return DeltaPoint(*self.ipoints(step))

    def schedule_np(self, steps: int) -> NPPoints:
        "the `steps` points of a full run, most noisy first, without the trailing clean point"
        return self._points(np.linspace(1, 0, steps, endpoint=False))

    def schedule(self, steps: int) -> Sequence[Point]:
        return tuple(_rows(self.schedule_np(steps)))


@dataclass(frozen=True)
class ScheduleCommon(SkrampleSchedule):
    base_timesteps: int = 1000
    "training timesteps; negative flips the timestep axis (T=N is then the clean end)"

    def _timestep_column(self, t: np.ndarray) -> np.ndarray:
        return ((1 - t) if self.base_timesteps < 0 else t) * abs(self.base_timesteps)

    @functools.cached_property
    def all_points(self) -> NPPoints:
        count = abs(self.base_timesteps)
        return self.points_np(np.linspace(0, 1, count if count > 1 else 10_000))

    @abstractmethod
    def _sigmas_to_points(self, sigmas: NPSequence, alphas: NPSequence) -> NPPoints: ...


@dataclass(frozen=True)
class FixedSchedule(SkrampleSchedule):
    "piecewise-linear interpolation through a given table (plus the clean end point)"

    fixed_schedule: Sequence[Point] | np.ndarray
    sigma_space: SigmaSpace

    @classmethod
    def from_regular(cls, timesteps, regular_sigmas, sigma_space: SigmaSpace):
        return cls(np.stack([timesteps, *sigma_space.normalize(regular_sigmas)], axis=1), sigma_space)

    def _points(self, t):
        from scipy.interpolate import make_interp_spline

        table = np.concatenate([np.asarray(self.fixed_schedule, dtype=np.float64), [[0, 0, 1]]])
        return make_interp_spline(np.linspace(0, 1, len(table)), table, k=1, axis=0)(1 - t)

    @property
    def space(self) -> SigmaSpace:
        return self.sigma_space

    def __hash__(self) -> int:  # ndarray field: hash by content so the LRU caches work
        return hash((np.asarray(self.fixed_schedule, dtype=np.float64).tobytes(), self.sigma_space))

    def __eq__(self, other) -> bool:
        return (
            isinstance(other, FixedSchedule)
            and self.sigma_space == other.sigma_space
            and np.array_equal(np.asarray(self.fixed_schedule, dtype=np.float64), np.asarray(other.fixed_schedule, dtype=np.float64))
        )


# ---------------------------------------------------------------------------------------------------
# base schedules
# ---------------------------------------------------------------------------------------------------
def _beta_power_integral(lo: float, slope: float, t: np.ndarray, power: float) -> np.ndarray:
    "int_0^t (lo + slope*u)^power du"
    return ((lo + slope * t) ** (power + 1) - lo ** (power + 1)) / (slope * (power + 1))


@dataclass(frozen=True)
class Scaled(ScheduleCommon):
    "Stable-Diffusion style beta schedule in closed (continuous) form"

    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_scale: float = 2

    @property
    def space(self) -> SigmaSpace:
        return VariancePreserving()

    def continuous_alphas_cumprod(self, t: NPSequence) -> NPSequence:
        """exp(-T * int_0^t (beta + beta^2/2)) with beta(u) = (b0^(1/k) + (b1^(1/k) - b0^(1/k)) u)^k:
        the continuum limit of cumprod(1 - beta_i)."""
        k = self.beta_scale
        lo = self.beta_start ** (1 / k)
        slope = self.beta_end ** (1 / k) - lo
        if abs(slope) < 1e-8:
            flat = lo**k
            first, second = flat * t, (flat**2) * t
        else:
            first = _beta_power_integral(lo, slope, t, k)
            second = _beta_power_integral(lo, slope, t, 2 * k)
        return np.exp(-(abs(self.base_timesteps) * (first + second / 2)))

    def _points(self, t):
        acp = self.continuous_alphas_cumprod(t)
        with np.errstate(divide="ignore"):  # acp == 0 (zero terminal SNR) -> sigma = inf -> (1, 0)
            regular = np.sqrt((1 - acp) / acp)
        return np.stack([self._timestep_column(t), *self.space.normalize(regular)], 1)

    def _sigmas_to_points(self, sigmas, alphas):
        table = self.all_points
        return np.stack([np.interp(sigmas, table[:, 1], table[:, 0]), sigmas, alphas], axis=1)


@dataclass(frozen=True)
class ZSNR(Scaled):
    "Scaled, shifted and rescaled to zero terminal SNR (arXiv 2305.08891, Algorithm 1)"

    def continuous_alphas_cumprod(self, t):
        root = np.sqrt(super().continuous_alphas_cumprod(np.concatenate([[0], t, [1]])))
        head, tail = root[0].item(), root[-1].item()
        body = root[1:-1]
        body -= tail
        body *= head / (head - tail)
        return body**2


@dataclass(frozen=True)
class Linear(ScheduleCommon):
    "sigma falls linearly from sigma_start to 0"

    sigma_start: float = 1
    custom_space: SigmaSpace | None = None

    @property
    def space(self) -> SigmaSpace:
        if self.custom_space is not None:
            return self.custom_space
        return FlowMatching() if self.sigma_start <= 1 else VariancePreserving()

    def _points(self, t):
        return np.stack([self._timestep_column(t), *self.space.normalize(t * self.sigma_start)], axis=1)

    def _sigmas_to_points(self, sigmas, alphas):
        axis = (self.sigma_start - sigmas) if self.base_timesteps < 0 else sigmas
        return np.stack([axis * (abs(self.base_timesteps) / self.sigma_start), sigmas, alphas], axis=1)


# ---------------------------------------------------------------------------------------------------
# schedules that wrap other schedules
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class _PartialSchedule(SkrampleSchedule):
    base: SkrampleSchedule

    @property
    def space(self) -> SigmaSpace:
        return self.base.space


@dataclass(frozen=True)
class SubSchedule(_PartialSchedule):
    "replaces the sigmas of a base schedule"

    base: ScheduleCommon

    @property
    def all(self):
        return (self, self.base)

    @property
    def lowest(self) -> ScheduleCommon:
        return self.base

    @property
    def base_timesteps(self) -> int:
        return self.base.base_timesteps


class SubSigmas(SubSchedule):
    @functools.cached_property
    def _base_regular_0(self) -> float:
        return self.base.space.regularize(self.base.point_0.sigma).item()

    @functools.cached_property
    def _base_regular_1(self) -> float:
        return self.base.space.regularize(self.base.point_1.sigma).item()

    @abstractmethod
    def _sub_sigmas(self, t): ...

    def _points(self, t):
        return self.base._sigmas_to_points(*self.space.normalize(self._sub_sigmas(t)))


@dataclass(frozen=True)
class ScheduleModifier(_PartialSchedule):
    "re-spaces time before handing it to the base schedule"

    base: SkrampleSchedule

    @abstractmethod
    def _modify(self, t): ...

    def _points(self, t):
        return self.base._points(self._modify(t))

    @property
    def all_split(self):
        "(modifiers outermost first, sub-schedule or None, base schedule)"
        mods: list[ScheduleModifier] = [self]
        inner = self.base
        while isinstance(inner, ScheduleModifier):
            mods.append(inner)
            inner = inner.base
        sub = None
        if isinstance(inner, SubSchedule):
            sub, inner = inner, inner.base
        return mods, sub, inner

    @property
    def all(self):
        mods, sub, base = self.all_split
        return [*mods, *([sub] if sub is not None else []), base]

    @property
    def lowest(self):
        return self.all_split[2]

    @staticmethod
    def stack(modifiers, sub, base):
        "inverse of all_split"
        built = base
        if sub is not None:
            assert isinstance(base, ScheduleCommon)
            built = replace(sub, base=built)
        for mod in reversed(modifiers):
            built = replace(mod, base=built)
        return built

    @staticmethod
    def _matches(obj, kind, exact: bool) -> bool:
        return type(obj) is kind or (not exact and isinstance(obj, kind))

    def find(self, skrample_schedule, exact: bool = False):
        for mod in self.all_split[0]:
            if self._matches(mod, skrample_schedule, exact):
                return mod
        return None

    def find_split(self, skrample_schedule, exact: bool = False):
        mods, sub, base = self.all_split
        hit, before, after = None, [], []
        for mod in mods:
            if self._matches(mod, skrample_schedule, exact):
                hit = mod
            elif hit is None:
                before.append(mod)
            else:
                after.append(mod)
        return (before, hit, after, sub, base) if hit else None


@dataclass(frozen=True)
class NoSub(SubSchedule):
    def _points(self, t):
        return self.base._points(t)


@dataclass(frozen=True)
class NoMod(ScheduleModifier):
    def _modify(self, t):
        return t


def _rescale(values: np.ndarray, top, bottom=0):
    return (values - bottom) / (top - bottom)


@dataclass(frozen=True)
class Karras(SubSigmas):
    "rho-ramp between the sigma of the 1/steps point and the maximum sigma (Karras et al. 2022)"

    rho: float = 7.0
    steps: float = 20

    @functools.cached_property
    def _base_regular_s(self) -> float:
        return self.base.space.regularize(self.base.point(1 / self.steps).sigma).item()

    def _sub_sigmas(self, t):
        lo, hi = self._base_regular_s ** (1.0 / self.rho), self._base_regular_1 ** (1.0 / self.rho)
        ramp = np.concatenate([[1, 0], t])
        sig = (lo * (1 - ramp) + hi * ramp) ** self.rho
        return _rescale(sig[2:], sig[0], sig[1]) * self._base_regular_1


@dataclass(frozen=True)
class Exponential(SubSigmas):
    "log-linear ('polyexponential' for rho != 1) ramp"

    rho: float = 1.0
    steps: float = 20

    @functools.cached_property
    def _base_regular_s(self) -> float:
        return self.base.space.regularize(self.base.point(1 / self.steps).sigma).item()

    def _sub_sigmas(self, t):
        ramp = np.concatenate([[1, 0], t]) ** self.rho
        sig = np.exp(np.log(self._base_regular_s) * (1 - ramp) + np.log(self._base_regular_1) * ramp)
        return _rescale(sig[2:], sig[0], sig[1]) * self._base_regular_1


@dataclass(frozen=True)
class Beta(SubSigmas):
    "Beta-distribution quantiles (arXiv 2407.12173)"

    alpha: float = 0.6
    beta: float = 0.6

    def _sub_sigmas(self, t):
        from scipy.stats import beta as beta_dist

        q = beta_dist.ppf(np.concatenate([[1], t]), self.alpha, self.beta)
        return _rescale(q, q[0])[1:] * self._base_regular_1


@dataclass(frozen=True)
class Probit(SubSigmas):
    "sigmoid of normal quantiles"

    scale: float = 3

    def _sub_sigmas(self, t):
        from scipy.stats import norm

        prob = np.concatenate([[1, 0], t]) * (1 - 1e-8)
        e = math.e ** norm.ppf(prob, scale=self.scale)
        s = e / (1 + e)
        return _rescale(s[2:], *s[:2]) * self._base_regular_1


@dataclass(frozen=True)
class FlowShift(ScheduleModifier):
    shift: float = 3.0

    def _modify(self, t):
        return self.shift * t / (1 + (self.shift - 1) * t)


@dataclass(frozen=True)
class Hyper(ScheduleModifier):
    "tanh (scale > 0) / sinh (scale < 0) time warp"

    scale: float = 2
    tail: bool = True

    def _modify(self, t):
        if abs(self.scale) <= 1e-8:
            return t
        bottom = -self.scale * self.tail
        p = np.concatenate([[1], t]) * (self.scale - bottom) + bottom
        p = np.sinh(p) if self.scale < 0 else np.tanh(p / math.sqrt(2))
        return _rescale(p[1:], p[0], -p[0] * self.tail)


@dataclass(frozen=True)
class Sinner(ScheduleModifier):
    "sine-wave time warp: y = sin(x) + x*k stays monotone for k >= 1"

    count: float = -2
    scale: float = 2

    def _modify(self, t):
        if abs(self.scale) <= 1e-8 or self.count == math.inf:
            return t
        half_cycles = rescale_positive(self.count * 2 ** math.copysign(1, self.count)) + 1
        phase = np.concatenate([[0, 1], 1 - t]) * (math.pi * half_cycles)
        if self.scale >= 0:
            phase += math.pi
        lift = abs(self.scale) ** -1 + 1
        wave = np.sin(phase) + phase * lift
        return _rescale(wave[2:], *wave[:2])
