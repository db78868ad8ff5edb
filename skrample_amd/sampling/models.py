"""Prediction spaces (what the network outputs) and the per-step scalars Gamma, Delta, zeta.

Same public classes as reference `skrample/sampling/models.py` (DiffusionModel :10-83, DataModel
:86-106, NoiseModel :109-128, FlowModel :131-152, VelocityModel :155-176, ScaleX :184-212,
ModelConvert :215-239).  Formulated here as *coefficient tables*: every model maps to two pairs of
fp64 scalars,

    x_hat = xs*sample + xo*output          (`x_weights`,   reference to_x)
    output = os*sample + ox*x_hat          (`out_weights`, reference from_x)

so conversions between spaces and the update  sample*Gamma + output*Delta + noise*zeta  compose into
plain coefficient arithmetic that the fused kernel applies in one pass (the reference runs each
`-`, `*`, `/` as its own full-tensor aten kernel).
"""

from __future__ import annotations

import abc
import dataclasses
import math
from functools import wraps
from typing import Callable

from ..common import DeltaPoint, Point
from .lazy import lift, settle


def _native():
    from . import native  # (native imports this module)

    return native


@dataclasses.dataclass(frozen=True)
class DiffusionModel(abc.ABC):
    # ---- linear coefficient tables ------------------------------------------------------------------
    @abc.abstractmethod
    def x_weights(self, point: Point) -> tuple[float, float]:
        "(xs, xo): x_hat = xs*sample + xo*output"

    @abc.abstractmethod
    def out_weights(self, point: Point) -> tuple[float, float]:
        "(os, ox): output = os*sample + ox*x_hat"

    @abc.abstractmethod
    def gamma(self, delta_point: DeltaPoint, eta: float = 0) -> float:
        "weight of the current sample in the update"

    @abc.abstractmethod
    def delta(self, delta_point: DeltaPoint, eta: float = 0) -> float:
        "weight of the model output in the update"

    # ---- op-by-op programs for the rounded conversion kernel (lazy.RoundedConversion) ------------------
    def to_x_program(self, point: Point) -> tuple[int, float, float] | None:
        "(kind, k0, k1) of include/skrample_hip.h `convert_to`, or None if this model has no program"
        return None

    def from_x_program(self, point: Point) -> tuple[int, float, float] | None:
        "(kind, k2, k3) of `convert_from`"
        return None

    # ---- stochastic part (shared by all spaces; reference models.py:30-51) ---------------------------
    def zeta_ts(self, delta: DeltaPoint, eta: float = 1.0, epsilon: float = 1e-8) -> float:
        "eta * std of the noise that the transition p(x_to | x_from) can absorb"
        src, dst = delta
        if abs(eta) < epsilon or abs(dst.sigma) < epsilon:
            return 0
        snr_ratio = (src.alpha * dst.sigma) / (dst.alpha * src.sigma)
        return eta * math.sqrt(max(0.0, (dst.sigma**2) * (1.0 - snr_ratio**2)))

    def zeta(self, delta_point: DeltaPoint, eta: float = 1.0) -> float:
        return self.zeta_ts(delta_point, eta)

    def eta_transform(self, delta_point: DeltaPoint, eta: float = 0) -> DeltaPoint:
        "shrink the destination sigma by the variance the injected noise will supply"
        src, dst = delta_point
        z = self.zeta_ts(delta_point, eta)
        if z != 0:
            dst = Point(dst.timestep, math.sqrt(max(0.0, dst.sigma**2 - z**2)), dst.alpha)
        return DeltaPoint(src, dst)

    # ---- value-level API (numbers, lazy forms or HIP tensors) ------------------------------------------
    def to_x(self, sample, output, point: Point):
        "output -> x_hat"
        done = _native().try_expr(lambda s_, o_: _native()._to_x(self, s_, o_, point), sample, output)  # 16-bit tensors: the reference's rounded ops
        if done is not None:
            return done
        xs, xo = self.x_weights(point)
        if xs == 0 and xo == 1:
            return output
        return settle(lift(sample) * xs + lift(output) * xo, like=output)

    def from_x(self, sample, x, point: Point):
        "x_hat -> output"
        done = _native().try_expr(lambda s_, x_: _native()._from_x(self, s_, x_, point), sample, x)
        if done is not None:
            return done
        os_, ox = self.out_weights(point)
        if os_ == 0 and ox == 1:
            return x
        return settle(lift(sample) * os_ + lift(x) * ox, like=x)

    def update_form(self, sample, output, delta_point: DeltaPoint, noise=None, eta: float = 0):
        "lazy  sample*Gamma + output*Delta (+ noise*zeta)"
        form = lift(sample) * self.gamma(delta_point, eta) + lift(output) * self.delta(delta_point, eta)
        if noise is not None:
            z = self.zeta(delta_point, eta)
            if z != 0:
                form = form + lift(noise) * z
        return form

    def forward(self, sample, output, delta_point: DeltaPoint, noise=None, eta: float = 0):
        "sample*Gamma + output*Delta + noise*zeta (reference models.py:53-67), one fused launch"
        operands = (sample, output) if noise is None else (sample, output, noise)
        done = _native().try_expr(lambda s_, o_, n_=None: _native()._forward(self, s_, o_, delta_point, n_, eta), *operands)
        if done is not None:
            return done
        return settle(self.update_form(sample, output, delta_point, noise, eta), like=sample)

    def backward(self, sample, result, delta_point: DeltaPoint, noise=None, eta: float = 0):
        "solve forward() for the output (reference models.py:69-83)"
        operands = (sample, result) if noise is None else (sample, result, noise)
        done = _native().try_expr(lambda s_, r_, n_=None: _native()._backward(self, s_, r_, delta_point, n_, eta), *operands)
        if done is not None:
            return done
        form = lift(result) - lift(sample) * self.gamma(delta_point, eta)
        if noise is not None:
            z = self.zeta(delta_point, eta)
            if z != 0:
                form = form - lift(noise) * z
        return settle(form / self.delta(delta_point, eta), like=sample)


def _reciprocal(x: float) -> float:
    """1/x.  A zero denominator (e.g. eps-prediction at alpha = 0) raises ZeroDivisionError exactly as the
    reference's scalar path does (models.py:117); the reference's tensor path would silently fill the
    latent with inf/nan instead -- here that failure is loud in both cases."""
    return 1.0 / x


@dataclasses.dataclass(frozen=True)
class DataModel(DiffusionModel):
    "x-prediction: the network predicts the clean sample"

    def to_x_program(self, point):
        return 0, 0.0, 0.0

    def from_x_program(self, point):
        return 0, 0.0, 0.0

    def x_weights(self, point):
        return 0.0, 1.0

    def out_weights(self, point):
        return 0.0, 1.0

    def gamma(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return dst.sigma / src.sigma

    def delta(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return dst.alpha - src.alpha * dst.sigma / src.sigma


@dataclasses.dataclass(frozen=True)
class NoiseModel(DiffusionModel):
    "epsilon-prediction: the network predicts the added noise"

    def to_x_program(self, point):
        return 1, point.sigma, point.alpha  # (s - sigma*o) / alpha

    def from_x_program(self, point):
        return 1, point.alpha, point.sigma  # (s - alpha*x) / sigma

    def x_weights(self, point):
        inv = _reciprocal(point.alpha)
        return inv, -point.sigma * inv

    def out_weights(self, point):
        inv = _reciprocal(point.sigma)
        return inv, -point.alpha * inv

    def gamma(self, delta_point, eta=0):
        return delta_point.point_to.alpha / delta_point.point_from.alpha

    def delta(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return dst.sigma - (dst.alpha * src.sigma) / src.alpha


@dataclasses.dataclass(frozen=True)
class FlowModel(DiffusionModel):
    "u-prediction (rectified flow / flow matching: FLUX, SD3)"

    def to_x_program(self, point):
        return 1, point.sigma, point.alpha + point.sigma  # (s - sigma*o) / (alpha + sigma)

    def from_x_program(self, point):
        return 1, point.alpha + point.sigma, point.sigma  # (s - (alpha+sigma)*x) / sigma

    def x_weights(self, point):
        inv = _reciprocal(point.alpha + point.sigma)
        return inv, -point.sigma * inv

    def out_weights(self, point):
        inv = _reciprocal(point.sigma)
        return inv, -(point.alpha + point.sigma) * inv

    def gamma(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return (dst.sigma + dst.alpha) / (src.sigma + src.alpha)

    def delta(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return (src.alpha * dst.sigma - dst.alpha * src.sigma) / (src.alpha + src.sigma)


@dataclasses.dataclass(frozen=True)
class VelocityModel(DiffusionModel):
    "v-prediction (zero-terminal-SNR models)"

    def to_x_program(self, point):
        return 2, point.sigma, point.alpha  # alpha*s - sigma*o

    def from_x_program(self, point):
        return 2, point.alpha, point.sigma  # (alpha*s - x) / sigma

    def x_weights(self, point):
        return point.alpha, -point.sigma

    def out_weights(self, point):
        inv = _reciprocal(point.sigma)
        return point.alpha * inv, -inv

    def gamma(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return (dst.sigma / src.sigma) * (1 - src.alpha * src.alpha) + dst.alpha * src.alpha

    def delta(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return src.alpha * dst.sigma - dst.alpha * src.sigma


@dataclasses.dataclass(frozen=True)
class FakeModel(DiffusionModel):
    "marker: a space used only as an alternative derivative space, never a network's output"


@dataclasses.dataclass(frozen=True)
class ScaleX(FakeModel):
    "x-prediction scaled by exp(-log10(|bias|+1) * (alpha or sigma))"

    bias: float = 3

    def x_scale(self, point: Point) -> float:
        return math.exp(-math.log10(abs(self.bias) + 1) * (point.sigma if self.bias < 0 else point.alpha))

    def to_x_program(self, point):
        return 3, self.x_scale(point), 0.0  # o * scale

    def from_x_program(self, point):
        return 3, self.x_scale(point), 0.0  # x / scale

    def x_weights(self, point):
        return 0.0, self.x_scale(point)

    def out_weights(self, point):
        return 0.0, 1.0 / self.x_scale(point)

    def to_x(self, sample, output, point):
        done = _native().try_expr(lambda o_: _native()._to_x(self, None, o_, point), output)
        return done if done is not None else settle(lift(output) * self.x_scale(point), like=output)

    def from_x(self, sample, x, point):
        done = _native().try_expr(lambda x_: _native()._from_x(self, None, x_, point), x)
        return done if done is not None else settle(lift(x) / self.x_scale(point), like=x)

    def gamma(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return dst.sigma / src.sigma

    def delta(self, delta_point, eta=0):
        src, dst = self.eta_transform(delta_point, eta)
        return (dst.alpha - src.alpha * dst.sigma / src.sigma) * self.x_scale(src)


@dataclasses.dataclass(frozen=True)
class ModelConvert:
    "re-express a network output of space `transform_from` in space `transform_to`"

    transform_from: DiffusionModel
    transform_to: DiffusionModel

    def weights_to(self, point: Point) -> tuple[float, float]:
        "(ws, wo): output_to = ws*sample + wo*output_from"
        if self.transform_to is self.transform_from:
            return 0.0, 1.0
        xs, xo = self.transform_from.x_weights(point)
        os_, ox = self.transform_to.out_weights(point)
        return os_ + ox * xs, ox * xo

    def weights_from(self, point: Point) -> tuple[float, float]:
        if self.transform_to is self.transform_from:
            return 0.0, 1.0
        xs, xo = self.transform_to.x_weights(point)
        os_, ox = self.transform_from.out_weights(point)
        return os_ + ox * xs, ox * xo

    def rounded_program(self, point: Point, negate_output: bool = False) -> tuple[int, int, list[float]] | None:
        """(to_kind, from_kind, [k0..k3]) reproducing output_to() one rounded tensor op at a time, or None when
        either model lacks a program.  `negate_output` folds a preceding `-output` (exact) into the program."""
        if self.transform_to is self.transform_from:
            return None
        a, b = self.transform_from.to_x_program(point), self.transform_to.from_x_program(point)
        if a is None or b is None:
            return None
        to_kind, k0, k1 = a
        if negate_output:
            if to_kind == 0:
                to_kind, k0 = 3, -1.0  # x = o * (-1)
            else:
                k0 = -k0
        return to_kind, b[0], [k0, k1, b[1], b[2]]

    def form_to(self, sample, output_from, point: Point):
        "lazy conversion (no launch)"
        ws, wo = self.weights_to(point)
        if ws == 0 and wo == 1:
            return lift(output_from)
        if ws == 0:
            return lift(output_from) * wo
        return lift(sample) * ws + lift(output_from) * wo

    def output_to(self, sample, output_from, point: Point):
        if self.transform_to is self.transform_from:
            return output_from
        done = _native().try_expr(lambda s_, o_: _native()._output_to(self.transform_from, self.transform_to, s_, o_, point), sample, output_from)
        if done is not None:
            return done
        return settle(self.form_to(sample, output_from, point), like=output_from)

    def output_from(self, sample, output_to, point: Point):
        if self.transform_from is self.transform_to:
            return output_to
        done = _native().try_expr(lambda s_, o_: _native()._output_to(self.transform_to, self.transform_from, s_, o_, point), sample, output_to)
        if done is not None:
            return done
        ws, wo = self.weights_from(point)
        return settle(lift(sample) * ws + lift(output_to) * wo, like=output_to)

    def wrap_model_call(self, model: Callable) -> Callable:
        @wraps(model)
        def converted(x, t: float, s: float, a: float):
            return self.output_to(x, model(x, t, s, a), Point(t, s, a))

        return converted
