"""Scalar helpers.  Follows reference skrample/common.py."""

from __future__ import annotations

import math
from fractions import Fraction
from functools import lru_cache
from typing import NamedTuple

import numpy as np


class Pt(NamedTuple):
    """(timestep, sigma, alpha) -- reference common.py:24-30 `Point`."""

    t: float
    s: float
    a: float


def pt_add_noise(p: Pt, sample, noise):
    "common.py:32-33"
    return sample * p.a + noise * p.s


def pt_remove_noise(p: Pt, sample, noise):
    "common.py:35-40 (alpha == 0 on python floats raises ZeroDivisionError -> returns the scaled noise)"
    scaled = noise * p.s
    try:
        return (sample - scaled) / p.a
    except ZeroDivisionError:
        return scaled


class Stp(NamedTuple):
    """Normalised step (time_from, time_to) in 0..1 -- common.py:55-97 `Step`."""

    t0: float
    t1: float


def stp_from_int(position: int, amount: int) -> Stp:
    "common.py:66-68"
    return Stp(position / amount, (position + 1) / amount)


def stp_distance(s) -> float:
    return s[1] - s[0]


def stp_position(s) -> float:
    "common.py:85-88"
    return s[0] / stp_distance(s)


def stp_amount(s) -> float:
    "common.py:90-93"
    return 1 / stp_distance(s)


def stp_normal(s) -> Stp:
    "common.py:95-97"
    return Stp(min(s), max(s))


def stp_clamp(s) -> Stp:
    "common.py:80-83"
    d = stp_distance(s)
    return Stp(max(0, min(1 - d, s[0])), max(d, min(1, s[1])))


def divf(lhs: float, rhs: float) -> float:
    "common.py:133-140: float division with signed infinity; 0/0 raises"
    if rhs != 0:
        return lhs / rhs
    if lhs == 0:
        raise ZeroDivisionError
    return math.copysign(math.inf, lhs)


def ln(x: float) -> float:
    "common.py:143-150"
    if x > 0:
        return math.log(x)
    if x < 0:
        raise ValueError
    return -math.inf


def rescale_positive(x: float) -> float:
    "common.py:163-165"
    return (abs(x) + 1) ** math.copysign(1, x)


def softmax2(a: float, b: float) -> tuple[float, float]:
    "common.py:173-184 restricted to the 2-tuple the samplers use (math.e ** x, not math.exp)"
    ea, eb = math.e**a, math.e**b
    return ea / (ea + eb), eb / (ea + eb)


def spowf(x, f: float):
    "common.py:187-190"
    return abs(x) ** f * (-1 * (x < 0) | 1)


@lru_cache(maxsize=None)
def bashforth(order: int) -> tuple[float, ...]:
    "common.py:205-213: Adams-Bashforth weights from the Vandermonde system"
    m = [[(-j) ** k for j in range(order)] for k in range(order)]
    rhs = [1 / (k + 1) for k in range(order)]
    return tuple(np.linalg.solve(m, rhs).tolist())


def sumprod(p, q):
    """CPython >= 3.12 ``math.sumprod`` semantics, which fix the rounding order of every tensor
    update in the reference (models.py:65,67; structured.py:319,426; functional.py:84,99):
      * all operands int/float  -> exactly-rounded dot product,
      * otherwise               -> total = 0; total = total + p_i * q_i   (left to right).
    """
    p, q = list(p), list(q)
    if len(p) != len(q):
        raise ValueError("Inputs are not the same length")
    scalar = (int, float)
    if all(type(v) in scalar for v in p) and all(type(v) in scalar for v in q):
        if all(math.isfinite(v) for v in (*p, *q)):
            return float(sum((Fraction(a) * Fraction(b) for a, b in zip(p, q)), Fraction(0)))
        return math.fsum(a * b for a, b in zip(p, q))
    total = 0
    for a, b in zip(p, q):
        total = total + a * b
    return total
