"""Prediction-space algebra (eps / v / flow / x  <->  x-hat-0) and the Gamma / Delta / zeta scalars.
Follows reference skrample/sampling/models.py.

A predictor is just a tag: "data", "eps", "flow", "v", or ("scalex", bias).
All tensor-valued functions keep the reference's operation order so fp32 results are bit-equal.
"""

from __future__ import annotations

import math

from .scalars import Pt, sumprod

Pred = "str | tuple[str, float]"


def _kind(pred) -> str:
    return pred if isinstance(pred, str) else pred[0]


def _xscale(pred, p: Pt) -> float:
    "models.py:192-196"
    bias = pred[1]
    return math.exp(-math.log10(abs(bias) + 1) * (p.s if bias < 0 else p.a))


def to_x(pred, sample, output, p: Pt):
    "prediction -> x-hat-0.  models.py:92-94, 115-117, 136-138, 160-162, 198-199"
    k = _kind(pred)
    if k == "data":
        return output
    if k == "eps":
        return (sample - p.s * output) / p.a
    if k == "flow":
        return (sample - p.s * output) / (p.a + p.s)
    if k == "v":
        return p.a * sample - p.s * output
    if k == "scalex":
        return output * _xscale(pred, p)
    raise KeyError(pred)


def from_x(pred, sample, x, p: Pt):
    "x-hat-0 -> prediction.  models.py:96-98, 119-121, 140-142, 164-166, 201-202"
    k = _kind(pred)
    if k == "data":
        return x
    if k == "eps":
        return (sample - p.a * x) / p.s
    if k == "flow":
        return (sample - (p.a + p.s) * x) / p.s
    if k == "v":
        return (p.a * sample - x) / p.s
    if k == "scalex":
        return x / _xscale(pred, p)
    raise KeyError(pred)


def zeta(p0: Pt, p1: Pt, eta: float = 1.0, epsilon: float = 1e-8) -> float:
    "models.py:30-42: conditional-variance noise weight"
    if abs(eta) < epsilon or abs(p1.s) < epsilon:
        return 0
    ratio = (p0.a * p1.s) / (p1.a * p0.s)
    return eta * math.sqrt(max(0.0, (p1.s**2) * (1.0 - ratio**2)))


def eta_shift(p0: Pt, p1: Pt, eta: float = 0) -> tuple[Pt, Pt]:
    "models.py:44-51: shrink the target sigma by the variance that the injected noise supplies"
    z = zeta(p0, p1, eta)
    if z != 0:
        p1 = Pt(p1.t, math.sqrt(max(0.0, p1.s**2 - z**2)), p1.a)
    return p0, p1


def gamma(pred, p0: Pt, p1: Pt, eta: float = 0) -> float:
    "models.py:100-102, 123-124, 144-146, 168-172, 204-206"
    k = _kind(pred)
    if k == "eps":
        return p1.a / p0.a
    f, t = eta_shift(p0, p1, eta)
    if k in ("data", "scalex"):
        return t.s / f.s
    if k == "flow":
        return (t.s + t.a) / (f.s + f.a)
    if k == "v":
        return (t.s / f.s) * (1 - f.a * f.a) + t.a * f.a
    raise KeyError(pred)


def delta(pred, p0: Pt, p1: Pt, eta: float = 0) -> float:
    "models.py:104-106, 126-128, 148-152, 174-176, 208-212"
    k = _kind(pred)
    f, t = eta_shift(p0, p1, eta)
    if k == "data":
        return t.a - f.a * t.s / f.s
    if k == "eps":
        return t.s - (t.a * f.s) / f.a
    if k == "flow":
        return (f.a * t.s - t.a * f.s) / (f.a + f.s)
    if k == "v":
        return f.a * t.s - t.a * f.s
    if k == "scalex":
        return (t.a - f.a * t.s / f.s) * _xscale(pred, f)
    raise KeyError(pred)


def forward(pred, sample, output, p0: Pt, p1: Pt, noise=None, eta: float = 0):
    "models.py:53-67: sample*Gamma + output*Delta (+ noise*zeta), accumulated by sumprod"
    g = gamma(pred, p0, p1, eta)
    d = delta(pred, p0, p1, eta)
    if noise is not None:
        z = zeta(p0, p1, eta)
        if z != 0:
            return sumprod((sample, output, noise), (g, d, z))
    return sumprod((sample, output), (g, d))


def backward(pred, sample, result, p0: Pt, p1: Pt, noise=None, eta: float = 0):
    "models.py:69-83"
    g = gamma(pred, p0, p1, eta)
    d = delta(pred, p0, p1, eta)
    if noise is not None:
        z = zeta(p0, p1, eta)
        if z != 0:
            return (result - sample * g - noise * z) / d
    return (result - sample * g) / d


def convert(pred_from, pred_to, sample, output, p: Pt, identical: bool = False):
    """models.py:220-224 ModelConvert.output_to.  The reference short-circuits only on object
    *identity* (`transform_to is transform_from`); two equal-but-distinct model objects still take
    the to_x/from_x round trip (exact for the data model, a last-bit rounding otherwise).  Tags have
    no identity, so callers pass `identical=True` where the reference hands the very same object down
    (UniPC -> unisolve, structured.py:484-497)."""
    if identical:
        return output
    return from_x(pred_to, sample, to_x(pred_from, sample, output, p), p)
