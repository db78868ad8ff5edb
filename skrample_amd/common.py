"""Host-side scalar primitives of the sampler-step engine.

Mirrors the public names of reference `skrample/common.py` (Point :24-40, DeltaPoint :43-52,
Step :55-97, MergeStrategy :100-130, divf/ln :133-150, bashforth :205-213 ...) so code written
against skrample imports unchanged.  Everything here is fp64 Python/numpy on the host: these values
become kernel coefficients, they never touch the latent tensor themselves.
"""

from __future__ import annotations

import enum
import math
from fractions import Fraction
from functools import lru_cache
from typing import Any, Callable, NamedTuple, Sequence, TypeVar

import numpy as np

T = TypeVar("T")
Sample = Any  # float | skrample_amd.sampling.lazy.Lin | torch.Tensor (HIP device)
RNG = Callable[["Step | None"], Any]


class Point(NamedTuple):
    "One location on a noise schedule (reference common.py:24-30)"

    timestep: float
    sigma: float
    alpha: float

    def add_noise(self, sample, noise):
        "alpha*sample + sigma*noise (reference common.py:32-33).  Tensors go through the HIP engine."
        from .sampling import native
        from .sampling.lazy import lift, settle

        done = native.try_point("add", self, sample, noise)  # 16-bit tensors: the reference's three rounded ops, one launch
        return done if done is not None else settle(lift(sample) * self.alpha + lift(noise) * self.sigma, like=sample)

    def remove_noise(self, sample, noise):
        "(sample - sigma*noise)/alpha; alpha == 0 returns the scaled noise (reference common.py:35-40)"
        from .sampling import native
        from .sampling.lazy import lift, settle

        done = native.try_point("remove", self, sample, noise)  # 16-bit tensors, and any tensor at alpha = 0 (inf / nan, as the reference's)
        if done is not None:
            return done
        if self.alpha == 0:
            return settle(lift(noise) * self.sigma, like=sample)
        return settle((lift(sample) - lift(noise) * self.sigma) / self.alpha, like=sample)


class DeltaPoint(NamedTuple):
    point_from: Point
    point_to: Point

    def difference(self) -> Point:
        a, b = self
        return Point(b.timestep - a.timestep, b.sigma - a.sigma, b.alpha - a.alpha)


def clamp(x: float, low: float = 0, high: float = 1) -> float:
    return max(low, min(high, x))


class Step(NamedTuple):
    """A sampling step as a pair of normalised times in 0..1 (0 = all noise, 1 = clean);
    reference common.py:55-97."""

    time_from: float
    time_to: float

    @staticmethod
    def from_int(position: int, amount: int) -> "Step":
        return Step(position / amount, (position + 1) / amount)

    def distance(self) -> float:
        return self.time_to - self.time_from

    def offset(self, steps: float) -> "Step":
        shift = self.distance() * steps
        return Step(self.time_from + shift, self.time_to + shift)

    def clamp(self) -> "Step":
        width = self.distance()
        return Step(clamp(self.time_from, high=1 - width), clamp(self.time_to, low=width))

    def position(self) -> float:
        return self.time_from / self.distance()

    def amount(self) -> float:
        return 1 / self.distance()

    def normal(self) -> "Step":
        return Step(min(self), max(self))


@enum.unique
class MergeStrategy(str, enum.Enum):
    "How two modifier lists are combined (reference common.py:100-130)"

    Ours = "ours"
    Theirs = "theirs"
    After = "after"
    Before = "before"
    UniqueAfter = "uniqueafter"
    UniqueBefore = "uniquebefore"

    def __str__(self) -> str:
        return str(self.value)

    def merge(self, ours: list, theirs: list, cmp: Callable[[Any, Any], bool] = lambda a, b: a == b) -> list:
        def missing_from(pool: list, items: list) -> list:
            return [i for i in items if not any(cmp(p, i) for p in pool)]

        if self is MergeStrategy.Ours:
            return ours
        if self is MergeStrategy.Theirs:
            return theirs
        if self is MergeStrategy.After:
            return ours + theirs
        if self is MergeStrategy.Before:
            return theirs + ours
        if self is MergeStrategy.UniqueAfter:
            return ours + missing_from(ours, theirs)
        return theirs + missing_from(theirs, ours)


def divf(lhs: float, rhs: float) -> float:
    "division that yields signed infinity for x/0 and refuses 0/0 (reference common.py:133-140)"
    if rhs != 0:
        return lhs / rhs
    if lhs == 0:
        raise ZeroDivisionError
    return math.copysign(math.inf, lhs)


def ln(x: float) -> float:
    "log with ln(0) = -inf (reference common.py:143-150)"
    if x > 0:
        return math.log(x)
    if x < 0:
        raise ValueError
    return -math.inf


def normalize(regular, start: float, end: float = 0):
    return (regular - end) / (start - end)


def regularize(normal, start: float, end: float = 0):
    return normal * (start - end) + end


def rescale_positive(x: float) -> float:
    return (abs(x) + 1) ** math.copysign(1, x)


def rescale_subnormal(x: float) -> float:
    return math.copysign(1 - (abs(x) + 1) ** -1, x)


def exp(x):
    return math.e**x


def sigmoid(array):
    e = exp(array)
    return e / (1 + e)


def softmax(elems: Sequence) -> tuple:
    weights = [exp(e) for e in elems]
    total = sum(weights)
    return tuple(w / total for w in weights)


def spowf(x, f: float):
    "sign-preserving power (reference common.py:187-190)"
    return abs(x) ** f * (-1 * (x < 0) | 1)


def mean(x) -> float:
    return x if isinstance(x, (float, int)) else x.mean().item()


@lru_cache(maxsize=None)
def bashforth(order: int) -> tuple[float, ...]:
    """Adams-Bashforth weights b_j of the `order`-step explicit method: the solution of
    sum_j b_j (-j)^k = 1/(k+1), k < order (reference common.py:205-213)."""
    vander = [[(-j) ** k for j in range(order)] for k in range(order)]
    moments = [1 / (k + 1) for k in range(order)]
    return tuple(np.linalg.solve(vander, moments).tolist())


def sumprod(p: Sequence, q: Sequence):
    """`math.sumprod` (Python >= 3.12) for this package's value types: an exactly rounded dot product
    for plain numbers, a left-to-right lazy combination otherwise."""
    p, q = list(p), list(q)
    if len(p) != len(q):
        raise ValueError("Inputs are not the same length")
    plain = (int, float)
    if all(type(v) in plain for v in p) and all(type(v) in plain for v in q):
        if all(math.isfinite(v) for v in (*p, *q)):
            return float(sum((Fraction(a) * Fraction(b) for a, b in zip(p, q)), Fraction(0)))
        return math.fsum(a * b for a, b in zip(p, q))
    total = 0
    for a, b in zip(p, q):
        total = total + a * b
    return total
