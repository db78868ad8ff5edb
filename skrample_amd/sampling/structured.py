"""Structured ("inside-out") samplers: one solver step per call, history owned by the caller.

Public surface = reference `skrample/sampling/structured.py`: SampleInput/SKSamples (:16-40),
StructuredSampler (:43-91), StructuredMultistep.effective_order (:128-149), Euler (:163-180),
DPM (:183-283), Adams (:286-330), UniP (:333-445), UniPC (:448-497), SPC (:500-577).

Implementation is different by design.  Each sampler reduces its step to *scalar weights* over the
x-hat history (`_history_weights`), the conversion of every history entry to x-hat is a pair of
scalars (models.ModelConvert.weights_to), and the update is Gamma/Delta/zeta -- so a whole step,
including predictor conversion, multistep correction and noise injection, is one lazy linear form
(`lazy.Lin`) that `lazy.evaluate` executes as ONE fused HIP kernel.  The reference performs the same
step as ~16 full-tensor aten passes plus ~17 copies.

History entries are kept as *aliases* of the caller's tensors (never copied, never re-materialised):
`SKSamples.sample/.prediction` are the input tensors themselves.  Callers must not overwrite them in
place while they are still within `require_previous` steps of use.
"""

from __future__ import annotations

import dataclasses
import functools
import math
from abc import ABC, abstractmethod
from dataclasses import dataclass, replace
from typing import Any, Sequence

import numpy as np
import torch

from .. import common
from ..common import DeltaPoint, Point, Step, divf, ln, softmax, spowf  # noqa: F401  (spowf re-exported, as in the reference)
from ..scheduling import SkrampleSchedule, ipoint_lru
from . import lazy, models, native, traits
from .lazy import LazyTensor, Lin, lift


def _ipoint(schedule: SkrampleSchedule, t: float) -> Point:
    try:
        return ipoint_lru(schedule, float(t))
    except TypeError:  # unhashable user schedule
        return schedule.ipoint(t)


@dataclass(frozen=True)
class SampleInput:
    "inputs of one step (reference structured.py:16-34)"

    sample: Any
    "what the model saw"
    prediction: Any
    "what the model returned"
    step: Step
    noise: Any
    "extra noise for stochastic samplers: a tensor, a `lazy.PhiloxNoise`, or None"

    def delta_point(self, schedule: SkrampleSchedule) -> DeltaPoint:
        return DeltaPoint(_ipoint(schedule, self.step[0]), _ipoint(schedule, self.step[1]))


@dataclass(frozen=True)
class SKSamples(SampleInput):
    final: Any
    "the step result"


def _result_dtype(value) -> torch.dtype | None:
    if isinstance(value, torch.Tensor):
        return value.dtype
    if isinstance(value, LazyTensor):
        return value.dtype
    if type(value).__module__ == "numpy" and hasattr(value, "__array_interface__") and getattr(value, "ndim", 0) > 0:
        return lazy._from_numpy(value).dtype  # ndarrays keep their width too (float64 state and predictions for float64 arrays)
    return None


def _state_dtype(result_dtype: torch.dtype | None) -> torch.dtype:
    "dtype of derived state that must survive to the next step (reference keeps it in compute_scale)"
    ctx = lazy._compute_dtype.get()
    if ctx is not None:
        return ctx
    return torch.float64 if result_dtype == torch.float64 else torch.float32


@dataclass(frozen=True)
class StructuredSampler(ABC, traits.SamplingCommon):
    @property
    def require_noise(self) -> bool:
        return False

    @property
    def require_previous(self) -> int:
        return 0

    @abstractmethod
    def sample_packed(self, packed: SampleInput, model_transform: models.DiffusionModel, schedule: SkrampleSchedule, previous: Sequence[SKSamples] = ()) -> SKSamples: ...

    def sample(self, sample, prediction, step, model_transform, schedule, noise=None, previous: Sequence[SKSamples] = ()) -> SKSamples:
        return self.sample_packed(SampleInput(sample, prediction, Step(*step), noise), model_transform, schedule, previous)

    def scale_input(self, sample, point: Point):
        return sample


@dataclass(frozen=True)
class StatedSampler(StructuredSampler):
    "samplers whose record is just the inputs + final"

    @abstractmethod
    def _form(self, packed: SampleInput, model_transform, schedule, previous):
        "the step result as a number or lazy form (no kernel launch)"

    def sample_packed(self, packed, model_transform, schedule, previous=()):
        # called on 16-bit device tensors directly (no compute_scale context): the reference's op-by-op arithmetic, replayed in one launch
        record = native.try_stated(self, packed, model_transform, schedule, previous)
        if record is not None:
            return record
        form = self._form(packed, model_transform, schedule, previous)
        final = lazy.settle(form, dtype=_result_dtype(packed.sample))
        return SKSamples(packed.sample, packed.prediction, packed.step, packed.noise, final)


@dataclass(frozen=True)
class StructuredMultistep(traits.HigherOrder, StructuredSampler):
    @property
    def require_previous(self) -> int:
        return max(min(self.order, self.max_order()), self.min_order()) - 1

    def effective_order(self, step: Step, previous: Sequence) -> int:
        "order usable at this step: limited by position from the start, history, and distance to the end"
        pos = step.position()
        return max(1, min(self.max_order(), round(pos + 1), self.order, len(previous) + 1, round(step.amount() - pos)))


@dataclass(frozen=True)
class StructuredStochastic(traits.Stochastic, StructuredSampler):
    @property
    def require_noise(self) -> bool:
        return abs(self.stochasticity) > 1e-8


@dataclass(frozen=True)
class StructuredUnified(traits.UnifiedModelling, StructuredStochastic, StructuredMultistep):
    def _xhat_history(self, packed: SampleInput, model_transform, schedule, previous, count: int):
        """[q_0 (current), q_1 (previous), ...] as lazy forms in the derivative space, plus that space.
        Each entry is `ws*sample + wo*prediction` of the *aliased* input tensors of its own step."""
        if self.derivative_transform:
            conv = models.ModelConvert(model_transform, self.derivative_transform)
            space = self.derivative_transform
            forms = [conv.form_to(packed.sample, packed.prediction, _ipoint(schedule, packed.step[0]))]
            for rec in list(reversed(previous))[: count - 1]:
                forms.append(conv.form_to(rec.sample, rec.prediction, _ipoint(schedule, rec.step[0])))
        else:
            space = model_transform
            forms = [lift(packed.prediction)] + [lift(rec.prediction) for rec in list(reversed(previous))[: count - 1]]
        return forms, space


def _half_log_snr(point: Point) -> float:
    return ln(divf(point.alpha, point.sigma))


@dataclass(frozen=True)
class Euler(StructuredStochastic, StatedSampler):
    "first order; with stochasticity this is Euler-Maruyama / ancestral sampling"

    def _form(self, packed, model_transform, schedule, previous):
        return model_transform.update_form(packed.sample, packed.prediction, packed.delta_point(schedule), packed.noise, self.stochasticity)


@dataclass(frozen=True)
class DPM(StructuredUnified, StatedSampler):
    "DPM-Solver++ multistep, orders 1-3 (arXiv 2211.01095), ODE or SDE"

    @staticmethod
    def max_order() -> int:
        return 3

    @staticmethod
    def _history_weights(order: int, lam: float, lam_next: float, lam_prev: float, lam_prev2: float | None) -> list[float]:
        """weights of (q_0, q_1, q_2) in the corrected data prediction.
        order 2:  q = q0 + (q0 - q1) / (2 r)                       r  = (lam - lam_prev)/h
        order 3:  q = q0 + c1*D1 + c2*D2 with the divided differences D1, D2 of (q0,q1,q2) and
                  c1 = (phi/(-h) - 1)/phi, c2 = ((phi + h)/h^2 - 1/2)/phi, phi = expm1(-h)."""
        if order < 2:
            return [1.0]
        h = abs(lam_next - lam)
        r = (lam - lam_prev) / h
        inv_r = 1.0 / r
        if order == 2:
            return [1.0 + 0.5 * inv_r, -0.5 * inv_r]
        r2 = (lam_prev - lam_prev2) / h
        inv_r2 = 1.0 / r2
        phi = math.expm1(-h)
        c1 = ((phi / -h) - 1.0) / phi if phi != 0 else 0
        c2 = ((phi + h) / h**2 - 0.5) / phi if phi != 0 else 0
        span = r + r2
        on_d10 = c1 * (1.0 + r / span) + c2 / span  # D1 = d10 + r/(r+r2) (d10 - d11), D2 = (d10 - d11)/(r+r2)
        on_d11 = -(c1 * r / span + c2 / span)
        return [1.0 + on_d10 * inv_r, -on_d10 * inv_r + on_d11 * inv_r2, -on_d11 * inv_r2]

    def _form(self, packed, model_transform, schedule, previous):
        delta = packed.delta_point(schedule)
        order = self.effective_order(packed.step, previous)
        forms, space = self._xhat_history(packed, model_transform, schedule, previous, order)
        lam_prev = _half_log_snr(_ipoint(schedule, previous[-1].step[0])) if order >= 2 else 0.0
        lam_prev2 = _half_log_snr(_ipoint(schedule, previous[-2].step[0])) if order >= 3 else None
        weights = self._history_weights(order, _half_log_snr(delta.point_from), _half_log_snr(delta.point_to), lam_prev, lam_prev2)
        q = forms[0] if order < 2 else sum((f * w for f, w in zip(forms[1:], weights[1:])), forms[0] * weights[0])
        return space.update_form(packed.sample, q, delta, packed.noise, self.stochasticity)


@dataclass(frozen=True)
class Adams(StructuredUnified, StatedSampler):
    "Adams-Bashforth weights on the x-hat history (diffusers' IPNDM at order 4)"

    @staticmethod
    def max_order() -> int:
        return 9

    def _form(self, packed, model_transform, schedule, previous):
        order = self.effective_order(packed.step, previous)
        delta = packed.delta_point(schedule)
        forms, space = self._xhat_history(packed, model_transform, schedule, previous, order)
        weights = common.bashforth(order)
        q = sum((f * w for f, w in zip(forms[1:order], weights[1:])), forms[0] * weights[0])
        return space.update_form(packed.sample, q, delta, packed.noise, self.stochasticity)


@functools.lru_cache(maxsize=8192)  # (a function of its numbers alone, and a 20 us linear solve; a run meets every step's system again in its next run)
def _unisolve_weights(fast_solve: bool, order: int, lam: float, lam_next: float, lam_history: tuple[float, ...], corrector: bool) -> tuple[tuple[float, ...], float]:
    """(weights over [q_0, q_1..q_{order-1}], weight of q_next).
    q = q0 + sum_k rho_k (q_k - q0)/r_k  [+ rho_c (q_next - q0)],  R rho = b with
    R_nk = r_k^(n-1), b_n = n! h_phi_n / B(h)."""
    h = abs(lam_next - lam)
    hh = -h
    phi_1 = math.expm1(hh)
    ratios: list[float] = []  # r_k as used in the linear system (non-finite -> 0)
    raw: list[float] = []  # r_k as used to divide the differences
    for lam_k in lam_history[: order - 1]:
        rk = (lam_k - lam) / h
        raw.append(rk)
        ratios.append(rk if math.isfinite(rk) else 0)
    if corrector:
        ratios.append(1.0)
    threshold = 1 if corrector else 2
    if not ratios or (order == threshold and fast_solve):
        rhos = [0.5]
    else:
        phi_k = phi_1 / hh - 1
        rows, rhs = [], []
        for n in range(1, len(ratios) + 1):
            rows.append([math.pow(v, n - 1) for v in ratios])
            rhs.append(phi_k * math.factorial(n) / phi_1)
            phi_k = phi_k / hh - 1 / math.factorial(n + 1)
        rhos = np.linalg.solve(rows, rhs).tolist()
    n_terms = len(raw) + (1 if corrector else 0)
    rhos = rhos[:n_terms]
    hist = [rho / rk for rho, rk in zip(rhos, raw)]  # rho/inf = 0: an infinitely distant point drops out
    w_next = rhos[len(raw)] if corrector and len(rhos) > len(raw) else 0.0
    return (1.0 - math.fsum(hist) - w_next, *hist), w_next


@dataclass(frozen=True)
class UniP(StructuredUnified, StatedSampler):
    "the UniPC predictor on its own (arXiv 2302.04867)"

    fast_solve: bool = False
    "use the closed-form rho = 1/2 for UniP-2 / UniC-1 instead of the linear solve"

    @staticmethod
    def max_order() -> int:
        return 9

    def _unisolve_weights(self, order: int, lam: float, lam_next: float, lam_history: Sequence[float], corrector: bool) -> tuple[list[float], float]:
        weights, w_next = _unisolve_weights(self.fast_solve, order, lam, lam_next, tuple(lam_history), corrector)
        return list(weights), w_next

    def _unisolve_form(self, packed, model_transform, schedule, previous, prediction_next=None):
        "UniP (prediction_next is None) or UniC (prediction_next given, already in derivative space)"
        delta = packed.delta_point(schedule)
        order = self.effective_order(packed.step, previous)
        forms, space = self._xhat_history(packed, model_transform, schedule, previous, order)
        lam_hist = [_half_log_snr(_ipoint(schedule, previous[-n].step[0])) for n in range(1, order)]
        weights, w_next = self._unisolve_weights(order, _half_log_snr(delta.point_from), _half_log_snr(delta.point_to), lam_hist, prediction_next is not None)
        q = sum((f * w for f, w in zip(forms[1:], weights[1:])), forms[0] * weights[0])
        if prediction_next is not None:
            if self.derivative_transform and model_transform is not self.derivative_transform:
                prediction_next = models.ModelConvert(model_transform, self.derivative_transform).form_to(packed.sample, prediction_next, delta.point_from)
            q = q + lift(prediction_next) * w_next
        return space.update_form(packed.sample, q, delta, packed.noise, self.stochasticity)

    def unisolve(self, packed, model_transform, schedule, previous, prediction_next=None):
        return lazy.settle(self._unisolve_form(packed, model_transform, schedule, previous, prediction_next), dtype=_result_dtype(packed.sample))

    def _form(self, packed, model_transform, schedule, previous):
        return self._unisolve_form(packed, model_transform, schedule, previous)


def _two_output_step(state_form, final_form, packed: SampleInput, prediction, result_dtype) -> SKSamples:
    "store `state_form` (the corrected/blended sample) and the step result with one launch"
    if isinstance(final_form, Lin):
        state, final = lazy.evaluate([state_form, final_form], [_state_dtype(result_dtype), result_dtype])
    else:
        state, final = state_form, final_form
    return SKSamples(state, prediction, packed.step, packed.noise, final)


def _as_prediction(form, dtype):
    "the converted prediction as a lazy tensor of the sample's dtype -- or, when its operands differ in dtype, of their promotion (what the reference's tensor ops return)"
    if not isinstance(form, Lin):
        return form
    seen = {leaf.dtype for leaf, c in form.expanded().terms.values() if isinstance(leaf, torch.Tensor) and c != 0.0}
    if len(seen) > 1 or (len(seen) == 1 and dtype is not None and next(iter(seen)) != dtype and lazy._compute_dtype.get() is None):
        promoted = None
        for d in seen:
            promoted = d if promoted is None else torch.promote_types(promoted, d)
        dtype = promoted
    return LazyTensor(form, dtype or torch.float32)


@dataclass(frozen=True)
class UniPC(UniP):
    """UniPC: correct the previous step with the new prediction (UniC), then predict from the
    corrected sample.  Corrector output (fp32 state) and predictor output come from the SAME kernel
    launch: out0 = corrected sample, out1 = chain*out0 + ...  (include/skrample_hip.h)."""

    predictor: StructuredSampler | None = None

    @staticmethod
    def max_order() -> int:
        return 9

    @property
    def require_noise(self) -> bool:
        return super().require_noise or (self.predictor.require_noise if self.predictor else False)

    @property
    def require_previous(self) -> int:
        return max(super().require_previous + 1, self.predictor.require_previous if self.predictor else 0)

    def sample_packed(self, packed, model_transform, schedule, previous=()):
        record = native.try_unipc(self, packed, model_transform, schedule, previous)  # (see StatedSampler.sample_packed)
        if record is not None:
            return record
        result_dtype = _result_dtype(packed.sample)
        if self.derivative_transform:
            conv = models.ModelConvert(model_transform, self.derivative_transform)
            q = conv.form_to(packed.sample, packed.prediction, _ipoint(schedule, packed.step[0]))
            space = self.derivative_transform
        else:
            q, space = lift(packed.prediction), model_transform
        prediction = _as_prediction(q, result_dtype)
        inner = replace(packed, prediction=q)
        state_form = None
        if previous:
            state_form = self._unisolve_form(previous[-1], space, schedule, previous[:-1], prediction_next=q)
            inner = replace(inner, sample=state_form.node() if isinstance(state_form, Lin) else state_form)
        stepper = self.predictor or self
        if isinstance(stepper, StatedSampler):
            final_form = stepper._unisolve_form(inner, space, schedule, previous) if stepper is self else stepper._form(inner, space, schedule, previous)
        else:  # composite predictor: let it run on the materialised corrected sample
            if isinstance(state_form, Lin):
                corrected = lazy.evaluate([state_form], [_state_dtype(result_dtype)])[0]
                inner = replace(inner, sample=corrected)
            rec = stepper.sample_packed(inner, space, schedule, previous)
            return replace(rec, prediction=prediction)
        if state_form is None:
            final = lazy.settle(final_form, dtype=result_dtype)
            return SKSamples(packed.sample, prediction, packed.step, packed.noise, final)
        return _two_output_step(state_form, final_form, packed, prediction, result_dtype)


@dataclass(frozen=True)
class SPC(traits.DerivativeTransform, StructuredSampler):
    """Simple predictor-corrector: re-run the previous step with a corrector sampler, blend it with
    the incoming sample (softmax of sigma/alpha), then predict.  The linear blend (power == 1) is part of
    the one fused step; other powers take three launches (corrector, signed-power blend, predictor)."""

    predictor: StructuredSampler = Euler()
    corrector: StructuredSampler = Adams(order=4)
    bias: float = 0
    power: float = 1
    adaptive: bool = True
    invert: bool = False

    @property
    def require_noise(self) -> bool:
        return self.predictor.require_noise or self.corrector.require_noise

    @property
    def require_previous(self) -> int:
        return max(self.predictor.require_previous, self.corrector.require_previous + 1)

    def _sample_packed_composed(self, packed, model_transform, schedule, previous):
        """The reference's own composition (structured.py:527-575) for tensors of one 16-bit dtype outside a compute scale and the linear blend: derivative
        conversion, corrector on the shifted history, `sample * p + corrected * c`, predictor -- every piece takes the tape (the reference's rounded ops), a
        launch apiece instead of one fused launch."""
        point_from = _ipoint(schedule, packed.step[0])
        if self.derivative_transform:
            convert = models.ModelConvert(model_transform, self.derivative_transform)
            packed = replace(packed, prediction=convert.output_to(packed.sample, packed.prediction, point_from))
            model_transform = convert.transform_to
        if previous:
            shifted = [replace(rec, prediction=nxt) for rec, nxt in zip(previous, (*(rec.prediction for rec in previous[1:]), packed.prediction))]
            corrected = self.corrector.sample_packed(shifted.pop(), model_transform, schedule, shifted).final
            wp, wc = (point_from.sigma, point_from.alpha) if self.adaptive else (0, 0)
            wp, wc = softmax((wp - self.bias, wc + self.bias))
            if self.invert:
                wp, wc = wc, wp
            blended = native.try_expr(lambda s_, c_: s_ * wp + c_ * wc, packed.sample, corrected)
            if blended is None:
                blended = lazy.settle(lift(packed.sample) * wp + lift(corrected) * wc, like=packed.sample)
            packed = replace(packed, sample=blended)
        return self.predictor.sample_packed(packed, model_transform, schedule, previous)

    def sample_packed(self, packed, model_transform, schedule, previous=()):
        if abs(self.power - 1) <= 1e-8 and native._eligible(packed.sample, packed.prediction):
            return self._sample_packed_composed(packed, model_transform, schedule, previous)
        result_dtype = _result_dtype(packed.sample)
        point_from = _ipoint(schedule, packed.step[0])
        if self.derivative_transform:
            q = models.ModelConvert(model_transform, self.derivative_transform).form_to(packed.sample, packed.prediction, point_from)
            space = self.derivative_transform
        else:
            q, space = lift(packed.prediction), model_transform
        prediction = _as_prediction(q, result_dtype)
        inner = replace(packed, prediction=q)
        state_form = None
        if previous:
            if not isinstance(self.corrector, StatedSampler) or not isinstance(self.predictor, StatedSampler):
                raise lazy.SkrampleHipError("SPC on this engine needs plain (Euler/DPM/Adams/UniP) predictor and corrector samplers")
            shifted = [replace(rec, prediction=nxt) for rec, nxt in zip(previous, (*(rec.prediction for rec in previous[1:]), q))]
            corrected = self.corrector._form(shifted[-1], space, schedule, shifted[:-1])
            wp, wc = (point_from.sigma, point_from.alpha) if self.adaptive else (0, 0)
            wp, wc = softmax((wp - self.bias, wc + self.bias))
            if self.invert:
                wp, wc = wc, wp
            if abs(self.power - 1) > 1e-8:
                if isinstance(corrected, Lin):
                    # the one non-linear tensor op of the samplers: materialise the corrector's result (one launch), blend
                    # (one elementwise launch), then predict from the blended sample as usual
                    wide = _state_dtype(result_dtype)  # compute_scale: float32, or float64 when asked for
                    if not isinstance(packed.sample, torch.Tensor):
                        # ndarrays: the reference's spowf multiplies by an int64 sign array (common.py:187-190), which makes numpy promote a
                        # float32 array to float64 -- the blended sample and everything computed from it are float64 there
                        wide = result_dtype = torch.float64
                    sample_t = packed.sample if isinstance(packed.sample, torch.Tensor) else lazy.settle(lift(packed.sample), dtype=wide)
                    blended = lazy.power_blend(sample_t, lazy.settle(corrected, dtype=wide), wp, wc, self.power, wide)
                    inner = replace(inner, sample=blended)
                    final = lazy.settle(self.predictor._form(inner, space, schedule, previous), dtype=result_dtype)
                    return SKSamples(blended, prediction, packed.step, packed.noise, final)
                blended = common.spowf(common.spowf(packed.sample, self.power) * wp + common.spowf(corrected, self.power) * wc, 1 / self.power)
            else:
                blended = lift(packed.sample) * wp + corrected * wc
            state_form = blended
            inner = replace(inner, sample=blended.node() if isinstance(blended, Lin) else blended)
        elif not isinstance(self.predictor, StatedSampler):
            return replace(self.predictor.sample_packed(inner, space, schedule, previous), prediction=prediction)
        final_form = self.predictor._form(inner, space, schedule, previous)
        if state_form is None:
            return SKSamples(packed.sample, prediction, packed.step, packed.noise, lazy.settle(final_form, dtype=result_dtype))
        return _two_output_step(state_form, final_form, packed, prediction, result_dtype)
