"""The reference's own arithmetic for samplers called on tensors directly: one rounded tensor operation at a time.

`StructuredSampler.sample(bf16_tensor, ...)` without a scheduler wrapper (or behind a wrapper with `compute_scale=None`) runs the
reference's generic code on torch tensors, so every `*`, `+`, `-`, `/` of the step is a torch op in the TENSOR dtype and rounds
(reference structured.py:70-86, 167-497; models.py:53-224).  The fused kernel (sampling/lazy.py -> skr_step_launch) collapses the
step into one linear form, accumulates it in fp32 and rounds once: closer to the exact answer, but 4-47 last-place units away from what
the reference returns on 16-bit tensors (tests/golden/native16.npz).

This module gives that entry point the reference's bits.  The step is *recorded*: the functions below perform the reference's
operations, in the reference's order, on `Val` handles whose operators append to a `Tape` instead of computing; the tape then runs as
ONE kernel (csrc/skr_tape.hip, `skr_tape_launch`) that keeps every intermediate in registers and rounds after each operation exactly as
the separate torch ops would -- one pass over HBM instead of the reference's ~16, same bits.

Which calls take this path (`mode`):
    "auto"    (default) tensors of ONE 16-bit dtype, outside any compute_scale context  -> tape (device tensors: one launch of
              skr_tape_launch; host tensors: the tape one torch op per entry, which is the reference itself); everything else -> fused form
    "always"  also fp32 / fp64 tensors (their fused result differs from the reference by fp32 / fp64 rounding only)
    "never"   always the fused kernel
The scheduler wrappers cast to `compute_scale` first (fp32 by default), as the reference's do (diffusers.py:575-599), and are not
affected unless compute_scale is None.
"""

from __future__ import annotations

import ctypes
import math
import os
import struct
from typing import Sequence

import numpy as np
import torch

from .. import _hip, common
from ..common import DeltaPoint, Point, divf, ln
from . import lazy, models

mode = "auto"
launches = 0  # tapes launched so far (tests assert that a call did / did not take this path)


class _Refused(Exception):
    "this step is outside what the tape covers: the caller takes the fused path"


class Tape:
    "operations of one step over `Val` handles; leaves are device tensors"

    def __init__(self, dtype: torch.dtype, shape, device, require_device: bool = True):
        self.dtype, self.shape, self.device, self.require_device = dtype, tuple(shape), device, require_device
        self.ops: list[tuple[int, int, int, float]] = []  # (code, a, b, k); the value an op defines is its position
        self.leaves: list[torch.Tensor] = []
        self._leaf_of: dict[int, Val] = {}

    def leaf(self, t) -> "Val":
        if isinstance(t, Val):
            return t
        if isinstance(t, lazy.PhiloxNoise):
            t = t.realize(self.dtype)
        if isinstance(t, lazy.LazyTensor):
            t = t.materialize()
        if not isinstance(t, torch.Tensor) or (self.require_device and not t.is_cuda) or t.dtype != self.dtype or tuple(t.shape) != self.shape or t.device != self.device or not t.is_contiguous():
            raise _Refused
        hit = self._leaf_of.get(id(t))
        if hit is None:
            if len(self.leaves) >= _hip.TAPE_MAX_INPUTS:
                raise _Refused
            self.leaves.append(t)
            hit = self._leaf_of[id(t)] = self._emit(_hip.TAPE_LOAD, len(self.leaves) - 1, 0, 0.0)
        return hit

    def _emit(self, code: int, a: int, b: int, k: float) -> "Val":
        self.ops.append((code, a, b, float(k)))
        return Val(self, len(self.ops) - 1)


class Val:
    "a tensor-valued intermediate of the recorded step; arithmetic appends to the tape (torch's operator semantics on a tensor of one dtype)"

    __slots__ = ("tape", "n")
    __array_ufunc__ = None  # numpy scalars defer to the reflected operators below

    def __init__(self, tape: Tape, n: int):
        self.tape, self.n = tape, n

    def _bin(self, other, tensor_code: int, scalar_code: int, scalar_k=lambda k: k):
        if isinstance(other, Val):
            return self.tape._emit(tensor_code, self.n, other.n, 0.0)
        if isinstance(other, (int, float, np.floating, np.integer)):
            return self.tape._emit(scalar_code, self.n, 0, scalar_k(float(other)))
        if isinstance(other, (torch.Tensor, lazy.LazyTensor, lazy.PhiloxNoise)):
            return self.tape._emit(tensor_code, self.n, self.tape.leaf(other).n, 0.0)
        return NotImplemented

    def __add__(self, o):
        return self._bin(o, _hip.TAPE_ADD, _hip.TAPE_ADD_S)

    __radd__ = __add__  # a + k == k + a, one rounding either way

    def __sub__(self, o):
        return self._bin(o, _hip.TAPE_SUB, _hip.TAPE_ADD_S, lambda k: -k)  # a - k == a + (-k) exactly

    def __rsub__(self, o):
        if isinstance(o, (int, float, np.floating, np.integer)):
            return self.tape._emit(_hip.TAPE_RSUB_S, self.n, 0, float(o))
        return self.tape.leaf(o) - self

    def __mul__(self, o):
        return self._bin(o, _hip.TAPE_MUL, _hip.TAPE_MUL_S)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._bin(o, _hip.TAPE_DIV, _hip.TAPE_DIV_S)

    def __rtruediv__(self, o):
        if isinstance(o, (int, float, np.floating, np.integer)):  # torch: Tensor.__rtruediv__ = reciprocal() * other, two rounded ops
            return self.tape._emit(_hip.TAPE_RDIV_S, self.n, 0, 1.0) * float(o)
        return self.tape.leaf(o) / self

    def __neg__(self):
        return self.tape._emit(_hip.TAPE_NEG, self.n, 0, 0.0)


def _sumprod(ps, qs):
    "math.sumprod on non-float operands (CPython 3.12): total = 0; total = total + p * q, left to right -- reference models.py:65-67"
    total = 0
    for p, q in zip(ps, qs):
        total = total + p * q
    return total


# ---- prediction spaces: reference models.py:86-212 (to_x / from_x, op for op) -------------------------------------------------------
def _to_x(model, sample, output, point: Point):
    _t, sigma, alpha = point
    kind = type(model)
    if kind is models.DataModel:
        return output
    if kind is models.NoiseModel:
        return (sample - sigma * output) / alpha
    if kind is models.FlowModel:
        return (sample - sigma * output) / (alpha + sigma)
    if kind is models.VelocityModel:
        return alpha * sample - sigma * output
    if kind is models.ScaleX:
        return output * model.x_scale(point)
    raise _Refused


def _from_x(model, sample, x, point: Point):
    _t, sigma, alpha = point
    kind = type(model)
    if kind is models.DataModel:
        return x
    if kind is models.NoiseModel:
        return (sample - alpha * x) / sigma
    if kind is models.FlowModel:
        return (sample - (alpha + sigma) * x) / sigma
    if kind is models.VelocityModel:
        return (alpha * sample - x) / sigma
    if kind is models.ScaleX:
        return x / model.x_scale(point)
    raise _Refused


def _output_to(src, dst, sample, output, point: Point):
    "ModelConvert.output_to (reference models.py:220-224): identity only for the SAME object"
    return output if dst is src else _from_x(dst, sample, _to_x(src, sample, output, point), point)


def _forward(model, sample, output, delta: DeltaPoint, noise, eta: float):
    "DiffusionModel.forward (reference models.py:53-67)"
    gamma, dlt = model.gamma(delta, eta), model.delta(delta, eta)
    if noise is not None:
        zeta = model.zeta(delta, eta)
        if zeta != 0:
            return _sumprod((sample, output, noise), (gamma, dlt, zeta))
    return _sumprod((sample, output), (gamma, dlt))


class _Rec:
    "a step's inputs as the recorder sees them (SampleInput with Val / tensor fields)"

    __slots__ = ("sample", "prediction", "step", "noise")

    def __init__(self, sample, prediction, step, noise):
        self.sample, self.prediction, self.step, self.noise = sample, prediction, step, noise

    def delta_point(self, schedule) -> DeltaPoint:
        from .structured import _ipoint

        return DeltaPoint(_ipoint(schedule, self.step[0]), _ipoint(schedule, self.step[1]))


def _derivative_history(sampler, packed: _Rec, model, schedule, previous: Sequence, order: int, delta: DeltaPoint):
    "the `predictions` list of DPM / Adams / UniP and the space they live in (reference structured.py:209-225, 304-317, 360-377)"
    # (the reference slices previous[-effective_order + 1:], which at order 1 is the whole history: entries past `order` are converted
    #  there too, but nothing reads them -- they are not recorded here)
    older = list(previous[len(previous) - (order - 1) :]) if order > 1 else []
    if sampler.derivative_transform:
        dst = sampler.derivative_transform
        preds = [_output_to(model, dst, packed.sample, packed.prediction, delta.point_from)]
        preds += reversed([_output_to(model, dst, p.sample, p.prediction, p.delta_point(schedule).point_from) for p in older])
        return preds, dst
    return [packed.prediction, *reversed([p.prediction for p in older])], model


def _euler(sampler, packed: _Rec, model, schedule, previous):
    return _forward(model, packed.sample, packed.prediction, packed.delta_point(schedule), packed.noise, sampler.stochasticity)  # structured.py:174-180


def _dpm(sampler, packed: _Rec, model, schedule, previous):
    "reference structured.py:195-283"
    from .structured import _ipoint

    delta = packed.delta_point(schedule)
    order = sampler.effective_order(packed.step, previous)
    predictions, model = _derivative_history(sampler, packed, model, schedule, previous, order, delta)
    prediction = predictions.pop(0)
    if order >= 2:
        (_t0, sigma_u, sigma_v), (_t1, sigma_u_next, sigma_v_next) = delta
        lam, lam_next = ln(divf(sigma_v, sigma_u)), ln(divf(sigma_v_next, sigma_u_next))
        h = abs(lam_next - lam)
        _tp, su_prev, sv_prev = _ipoint(schedule, previous[-1].step[0])
        lam_prev = ln(divf(sv_prev, su_prev))
        r = (lam - lam_prev) / h
        prediction_prev = predictions.pop(0)
        d1_0 = (1.0 / r) * (prediction - prediction_prev)
        if order >= 3:
            _tp2, su_prev2, sv_prev2 = _ipoint(schedule, previous[-2].step[0])
            r2 = (lam_prev - ln(divf(sv_prev2, su_prev2))) / h
            prediction_p2 = predictions.pop(0)
            d1_1 = (1.0 / r2) * (prediction_prev - prediction_p2)
            d1 = d1_0 + (r / (r + r2)) * (d1_0 - d1_1)
            d2 = (1.0 / (r + r2)) * (d1_0 - d1_1)
            hh = -h
            e = math.expm1(hh)
            c1 = (e / hh - 1.0) / e if e != 0 else 0
            c2 = ((e - hh) / hh**2 - 0.5) / e if e != 0 else 0
            prediction = prediction + c1 * d1 + c2 * d2
        else:
            prediction = prediction + 0.5 * d1_0
    return _forward(model, packed.sample, prediction, delta, packed.noise, sampler.stochasticity)


def _adams(sampler, packed: _Rec, model, schedule, previous):
    "reference structured.py:294-330"
    order = sampler.effective_order(packed.step, previous)
    delta = packed.delta_point(schedule)
    predictions, model = _derivative_history(sampler, packed, model, schedule, previous, order, delta)
    weighted = _sumprod(predictions[:order], common.bashforth(order))
    return _forward(model, packed.sample, weighted, delta, packed.noise, sampler.stochasticity)


def _unisolve(sampler, packed: _Rec, model, schedule, previous, prediction_next=None):
    "UniP.unisolve (reference structured.py:344-436); with `prediction_next` it is the corrector"
    delta = packed.delta_point(schedule)
    order = sampler.effective_order(packed.step, previous)
    src_model = model
    predictions, model = _derivative_history(sampler, packed, model, schedule, previous, order, delta)
    if sampler.derivative_transform and prediction_next is not None:
        prediction_next = _output_to(src_model, model, packed.sample, prediction_next, delta.point_from)
    prediction = predictions.pop(0)
    (_t0, sigma_u, sigma_v), (_t1, sigma_u_next, sigma_v_next) = delta
    lam, lam_next = ln(divf(sigma_v, sigma_u)), ln(divf(sigma_v_next, sigma_u_next))
    h = abs(lam_next - lam)
    hh = -h
    phi_1 = math.expm1(hh)
    rks: list[float] = []
    d1s = []
    for n in range(1, order):
        prediction_prev = predictions.pop(0)
        _tn, su_n, sv_n = previous[-n].delta_point(schedule).point_from
        rk = (ln(divf(sv_n, su_n)) - lam) / h
        rks.append(rk if math.isfinite(rk) else 0)
        d1s.append((prediction_prev - prediction) / rk)
    if prediction_next is not None:
        rks.append(1.0)
        order_check = 1
        d1s.append(prediction_next - prediction)
    else:
        order_check = 2
    if not rks or (order == order_check and sampler.fast_solve):
        rhos = [0.5]
    else:
        phi_k = phi_1 / hh - 1
        rows, rhs = [], []
        for n in range(1, len(rks) + 1):
            rows.append([math.pow(v, n - 1) for v in rks])
            rhs.append(phi_k * math.factorial(n) / phi_1)
            phi_k = phi_k / hh - 1 / math.factorial(n + 1)
        rhos = np.linalg.solve(rows, rhs).tolist()
    result = _sumprod(rhos[: len(d1s)], d1s)
    prediction = prediction + result
    return _forward(model, packed.sample, prediction, delta, packed.noise, sampler.stochasticity)


def _stated(sampler):
    "the recorder of a sampler whose record is (inputs, final): exact types only -- a subclass may compute something else"
    from . import structured as S

    return {S.Euler: _euler, S.DPM: _dpm, S.Adams: _adams, S.UniP: _unisolve}.get(type(sampler))


def _wrap(tape: Tape, rec) -> _Rec:
    "a history record (SKSamples) with its tensors as leaves"
    return _Rec(tape.leaf(rec.sample), tape.leaf(rec.prediction), rec.step, None if rec.noise is None else rec.noise)


def _noise_leaf(tape: Tape, noise):
    return None if noise is None else tape.leaf(noise)


# ---- execution ----------------------------------------------------------------------------------------------------------------------
_TWO_REGISTERS = (_hip.TAPE_ADD, _hip.TAPE_SUB, _hip.TAPE_MUL, _hip.TAPE_DIV, _hip.TAPE_ADD_MS, _hip.TAPE_SUB_MS, _hip.TAPE_RSUB_MS)
fuse = os.environ.get("SKR_TAPE_NO_FUSE") is None  # (tuning / test switch: the launch form op for op as recorded)


def _fused(recorded: list, stored) -> list:
    """The recorded ops with every product by a scalar that one sum or difference reads -- and nothing else, no result either -- folded into its reader:
    `total + p * k` -> ADD_MS, `s - o * k` -> SUB_MS, `o * k - s` -> RSUB_MS, `0 + p * k` -> MULZ_S (the first term of a sumprod; `+ 0` turns -0 into +0,
    so it stays).  Same operations, same roundings, one visit of the kernel's dispatch and one trip through its register file fewer per pair; positions
    keep their meaning (a folded product leaves a None behind)."""
    T = _hip
    uses = [0] * len(recorded)
    for code, a, b, _k in recorded:
        if code != T.TAPE_LOAD:
            uses[a] += 1
            if code in _TWO_REGISTERS:
                uses[b] += 1
    ops: list = list(recorded)

    def product(v: int):
        "(operand, scalar) of value v if it is such a product"
        op = ops[v]
        return (op[1], op[3]) if op is not None and op[0] == T.TAPE_MUL_S and uses[v] == 1 and v not in stored else None

    for i, (code, a, b, k) in enumerate(recorded):
        if code == T.TAPE_ADD or code == T.TAPE_SUB:
            for mine, other, folded in ((b, a, T.TAPE_ADD_MS if code == T.TAPE_ADD else T.TAPE_SUB_MS), (a, b, T.TAPE_ADD_MS if code == T.TAPE_ADD else T.TAPE_RSUB_MS)):
                found = product(mine)
                if found is not None:
                    ops[i], ops[mine] = (folded, other, found[0], found[1]), None
                    break
        elif code == T.TAPE_ADD_S and k == 0.0 and math.copysign(1.0, k) > 0:
            found = product(a)
            if found is not None:
                ops[i], ops[a] = (T.TAPE_MULZ_S, found[0], 0, found[1]), None
    return ops


def _allocate(tape: Tape, results: list) -> tuple[list[tuple[int, int, int, int, float]], list[int], dict[int, int]]:
    """Registers for the straight-line tape, one pass: a value holds its register from its definition to its last read.  Returns the skr_tape
    ops (code, dst, a, b, k), the leaves they load (indices into tape.leaves, in input order) and the output slot of every stored value."""
    stores: dict[int, int] = {}  # value -> output slot
    for v in results:
        if tape.ops[v.n][0] != _hip.TAPE_LOAD and v.n not in stores:
            stores[v.n] = len(stores)
    if not stores or len(stores) > _hip.TAPE_MAX_OUTPUTS:
        raise _Refused
    ops = _fused(tape.ops, stores) if fuse else list(tape.ops)
    n = len(ops)
    reads = [() if op is None or op[0] == _hip.TAPE_LOAD else ((op[1], op[2]) if op[0] in _TWO_REGISTERS else (op[1],)) for op in ops]
    last_read = [-1] * n
    for i, rd in enumerate(reads):
        for r in rd:
            last_read[r] = i
    # (a value nothing reads -- e.g. a difference the reference computes for a term of weight zero -- stays on the tape: it costs no
    #  memory traffic; only the LOAD of a leaf nothing reads is dropped)
    free = list(range(_hip.TAPE_REGS - 1, -1, -1))
    reg_of: dict[int, int] = {}
    out_ops: list[tuple[int, int, int, int, float]] = []
    used_leaves: list[int] = []
    for i, op in enumerate(ops):
        if op is None or (op[0] == _hip.TAPE_LOAD and last_read[i] < 0):
            continue
        code, a, _b, k = op
        srcs = [reg_of[r] for r in reads[i]]
        for r in set(reads[i]):  # an operand read here for the last time hands its register on (the kernel reads before it writes)
            if last_read[r] == i:
                free.append(reg_of.pop(r))
        if not free:
            raise _Refused
        dst = free.pop()
        if code == _hip.TAPE_LOAD:
            if a not in used_leaves:
                used_leaves.append(a)
            out_ops.append((code, dst, used_leaves.index(a), 0, 0.0))
        else:
            out_ops.append((code, dst, srcs[0], srcs[1] if len(srcs) > 1 else 0, k))
        if i in stores:
            out_ops.append((_hip.TAPE_STORE, 0, dst, stores[i], 0.0))
        if last_read[i] < 0:
            free.append(dst)
        else:
            reg_of[i] = dst
    if len(out_ops) > _hip.TAPE_MAX_OPS or not used_leaves:
        raise _Refused
    return out_ops, used_leaves, stores


def _compile(tape: Tape, results: list):
    "the launch form of a tape: (skr_tape, indices into tape.leaves of its inputs, output slot of every stored value)"
    out_ops, used_leaves, stores = _allocate(tape, results)
    # the whole skr_tape in one pack (field-by-field ctypes stores were half of this function's time): header, then (code, dst, a, b, k) per op
    flat = [len(out_ops), len(used_leaves), len(stores), _hip.DTYPE_CODE[tape.dtype]]
    for entry in out_ops:
        flat.extend(entry)
    c = _hip.TapeC.from_buffer_copy(struct.pack("<4i" + "4id" * len(out_ops), *flat).ljust(ctypes.sizeof(_hip.TapeC), b"\0"))
    return c, used_leaves, stores


def _launch(c, leaves: list[torch.Tensor], n_outputs: int) -> list[torch.Tensor]:
    first = leaves[0]
    outs = [lazy.empty_output(first.shape, first.dtype, first.device) for _ in range(n_outputs)]
    ins = (ctypes.c_void_p * len(leaves))(*[t.data_ptr() for t in leaves])
    ous = (ctypes.c_void_p * n_outputs)(*[t.data_ptr() for t in outs])
    _hip.check(_hip.load().skr_tape_launch(ctypes.byref(c), ins, ous, first.numel(), _hip.current_stream_ptr(first.device)), "skr_tape_launch")
    global launches
    launches += 1
    return outs


def _run(tape: Tape, results: list) -> list[torch.Tensor]:
    """Fill a skr_tape and launch it.  `results` are the values to return: a leaf comes back as the caller's own tensor, anything else
    is stored by the launch right behind its definition."""
    c, used_leaves, stores = _compile(tape, results)
    outs = _launch(c, [tape.leaves[i] for i in used_leaves], len(stores))
    return [tape.leaves[tape.ops[v.n][1]] if tape.ops[v.n][0] == _hip.TAPE_LOAD else outs[stores[v.n]] for v in results]


# ---- remembered steps ---------------------------------------------------------------------------------------------------------------
# Everything a sampler step's tape holds besides the tensors -- op order, register numbers, the scalars -- is a function of (sampler, model,
# schedule, step, the steps of the history, which optional tensors are present, dtype, which arguments are one and the same tensor).  Recording
# is 20-150 us of host work per step (profiles/r05_bench_tape.txt), more than the launch takes up to a few million elements: a step seen before
# binds today's tensors to the remembered skr_tape instead.  Configuration objects the reference leaves unhashable are not remembered.
_remembered: dict = {}
REMEMBER_AT_MOST = 4096  # (3 KiB apiece)
remembered_hits = 0  # (diagnostics / tests)


def _settled(t, dtype):
    "what Tape.leaf turns an argument into"
    if isinstance(t, lazy.PhiloxNoise):
        return t.realize(dtype)
    if isinstance(t, lazy.LazyTensor):
        return t.materialize()
    return t


def _step_key(kind: str, sampler, model, schedule, packed, previous, arguments: list):
    first_seen: dict[int, int] = {}
    same = tuple(-1 if t is None else first_seen.setdefault(id(t), i) for i, t in enumerate(arguments))
    key = (kind, sampler, model, schedule, tuple(packed.step), tuple(tuple(p.step) for p in previous), packed.sample.dtype, same)
    try:
        hash(key)
    except TypeError:
        return None
    return key


def _remembered_step(kind: str, recorder, sampler, packed, model, schedule, previous, arguments: list) -> list:
    """`recorder(...) -> (tape, results)` run through the launch, or -- for a step remembered from an earlier call -- today's `arguments` (the
    tensors the recorder may read, in a fixed order) bound to the skr_tape recorded then.  Device tensors only; returns the tensors of `results`."""
    global remembered_hits
    like = packed.sample
    key = entry = None
    if like.is_cuda:
        settled = [None if t is None else _settled(t, like.dtype) for t in arguments]  # (what the recorder's leaves would be: sameness is judged on these)
        key = _step_key(kind, sampler, model, schedule, packed, previous, settled)
        entry = _remembered.get(key) if key is not None else None
    if entry is not None:
        c, used, n_outputs, answer = entry
        # (Tape.leaf's own conditions, on every argument the recorder would have made a leaf of -- read by the tape or not)
        if all(t is None or (isinstance(t, torch.Tensor) and t.dtype == like.dtype and t.shape == like.shape and t.device == like.device and t.is_contiguous()) for t in settled):
            outs = _launch(c, [settled[i] for i in used], n_outputs)
            remembered_hits += 1
            return [outs[n] if stored else settled[n] for stored, n in answer]
    tape, results = recorder(sampler, packed, model, schedule, previous, require_device=False)
    if tape.device.type != "cuda":
        return _run_host(tape, results)
    c, used_leaves, stores = _compile(tape, results)
    outs = _launch(c, [tape.leaves[i] for i in used_leaves], len(stores))
    if key is not None:
        where = {id(t): i for i, t in reversed(list(enumerate(settled))) if t is not None}
        used = [where.get(id(tape.leaves[i])) for i in used_leaves]
        answer = [(False, where.get(id(tape.leaves[tape.ops[v.n][1]]))) if tape.ops[v.n][0] == _hip.TAPE_LOAD else (True, stores[v.n]) for v in results]
        if None not in used and all(n is not None for _stored, n in answer):
            if len(_remembered) >= REMEMBER_AT_MOST:
                _remembered.clear()
            _remembered[key] = (c, used, len(stores), answer)
    return [tape.leaves[tape.ops[v.n][1]] if tape.ops[v.n][0] == _hip.TAPE_LOAD else outs[stores[v.n]] for v in results]


def _run_host(tape: Tape, results: list) -> list[torch.Tensor]:
    "host-resident tensors (the reference's own habitat): the tape one torch op per entry -- which IS the reference's sequence of aten calls"
    vals: list[torch.Tensor] = []
    for code, a, b, k in tape.ops:
        x = tape.leaves[a] if code == _hip.TAPE_LOAD else vals[a]
        if code == _hip.TAPE_LOAD:
            y = x
        elif code == _hip.TAPE_MUL_S:
            y = x * k
        elif code == _hip.TAPE_DIV_S:
            y = x / k
        elif code == _hip.TAPE_ADD_S:
            y = x + k
        elif code == _hip.TAPE_RSUB_S:
            y = k - x
        elif code == _hip.TAPE_RDIV_S:
            y = x.reciprocal() if k == 1.0 else (torch.tensor(k, dtype=x.dtype if x.dtype == torch.float64 else torch.float32) / x.to(x.dtype if x.dtype == torch.float64 else torch.float32)).to(x.dtype)
        elif code == _hip.TAPE_ADD:
            y = x + vals[b]
        elif code == _hip.TAPE_SUB:
            y = x - vals[b]
        elif code == _hip.TAPE_MUL:
            y = x * vals[b]
        elif code == _hip.TAPE_DIV:
            y = x / vals[b]
        else:
            y = -x
        vals.append(y)
    return [vals[v.n] for v in results]


def _execute(tape: Tape, results: list) -> list[torch.Tensor]:
    return _run(tape, results) if tape.device.type == "cuda" else _run_host(tape, results)


def _eligible(sample, prediction) -> bool:
    if mode == "never" or lazy._compute_dtype.get() is not None:
        return False
    if not (isinstance(sample, torch.Tensor) and isinstance(prediction, torch.Tensor) and sample.device == prediction.device):
        return False
    if sample.dtype != prediction.dtype or sample.shape != prediction.shape:
        return False
    if sample.dtype in (torch.bfloat16, torch.float16):
        return True
    return mode == "always" and sample.dtype in (torch.float32, torch.float64)


def record_stated(sampler, packed, model, schedule, previous, require_device: bool = True):
    "(tape, [final]) of a StatedSampler step in the reference's arithmetic; raises _Refused outside the tape's coverage"
    recorder = _stated(sampler)
    if recorder is None:
        raise _Refused
    tape = Tape(packed.sample.dtype, packed.sample.shape, packed.sample.device, require_device)
    me = _Rec(tape.leaf(packed.sample), tape.leaf(packed.prediction), packed.step, _noise_leaf(tape, packed.noise))
    hist = [_wrap(tape, p) for p in previous]
    return tape, [recorder(sampler, me, model, schedule, hist)]


def record_unipc(sampler, packed, model, schedule, previous, require_device: bool = True):
    "(tape, [record sample, record prediction, final]) of UniPC.sample_packed (reference structured.py:469-497): conversion, corrector on the previous record, predictor"
    predictor = sampler.predictor
    if predictor is not None and _stated(predictor) is None:
        raise _Refused
    tape = Tape(packed.sample.dtype, packed.sample.shape, packed.sample.device, require_device)
    delta = _Rec(None, None, packed.step, None).delta_point(schedule)
    sample, prediction = tape.leaf(packed.sample), tape.leaf(packed.prediction)
    noise = _noise_leaf(tape, packed.noise)
    if sampler.derivative_transform:
        prediction = _output_to(model, sampler.derivative_transform, sample, prediction, delta.point_from)
        model = sampler.derivative_transform
    hist = [_wrap(tape, p) for p in previous]
    if hist:
        hist[-1].noise = _noise_leaf(tape, previous[-1].noise)  # the corrector re-applies the noise of the step it corrects
        sample = _unisolve(sampler, hist[-1], model, schedule, hist[:-1], prediction_next=prediction)
    me = _Rec(sample, prediction, packed.step, noise)
    final = _unisolve(sampler, me, model, schedule, hist) if predictor is None else _stated(predictor)(predictor, me, model, schedule, hist)
    return tape, [sample, prediction, final]


def try_stated(sampler, packed, model, schedule, previous):
    "StatedSampler.sample_packed in the reference's arithmetic, or None (the caller runs the fused form)"
    if _stated(sampler) is None or not _eligible(packed.sample, packed.prediction):
        return None
    from .structured import SKSamples

    arguments = [packed.sample, packed.prediction, packed.noise]
    for p in previous:
        arguments += (p.sample, p.prediction)
    try:
        (final,) = _remembered_step("stated", record_stated, sampler, packed, model, schedule, previous, arguments)
    except _Refused:
        return None
    return SKSamples(packed.sample, packed.prediction, packed.step, packed.noise, final)


def try_unipc(sampler, packed, model, schedule, previous):
    "UniPC.sample_packed in the reference's arithmetic (one tape: corrected sample, converted prediction and step result), or None"
    from . import structured as S

    if type(sampler) is not S.UniPC or not _eligible(packed.sample, packed.prediction):
        return None
    arguments = [packed.sample, packed.prediction, packed.noise]
    for p in previous:
        arguments += (p.sample, p.prediction)
    if previous:
        arguments.append(previous[-1].noise)
    try:
        sample, prediction, final = _remembered_step("unipc", record_unipc, sampler, packed, model, schedule, previous, arguments)
    except _Refused:
        return None
    return S.SKSamples(sample, prediction, packed.step, packed.noise, final)


def try_point(kind: str, point: Point, sample, noise):
    """`Point.add_noise` / `Point.remove_noise` on tensors in the reference's arithmetic (common.py:32-40: `sample * alpha + noise * sigma`,
    `(sample - noise * sigma) / alpha`, each operator a rounded tensor op), or None (the caller runs the fused form).  What a diffusers
    pipeline reaches through `scheduler.add_noise` / `scale_noise` on 16-bit latents.  A tensor divided by alpha = 0 is not an exception in
    the reference (only its float path catches ZeroDivisionError): the quotient is inf / nan, and so it is here, whatever `mode` says."""
    singular = kind == "remove" and point.alpha == 0
    if not (isinstance(sample, torch.Tensor) and isinstance(noise, torch.Tensor)):
        return None
    if singular:
        if sample.dtype != noise.dtype or sample.shape != noise.shape or sample.device != noise.device or sample.dtype not in (torch.bfloat16, torch.float16, torch.float32, torch.float64):
            return None
    elif not _eligible(sample, noise):
        return None
    try:
        tape = Tape(sample.dtype, sample.shape, sample.device, require_device=False)
        s, n = tape.leaf(sample), tape.leaf(noise)
        out = s * point.alpha + n * point.sigma if kind == "add" else (s - n * point.sigma) / point.alpha
        (done,) = _execute(tape, [out])
    except _Refused:
        return None
    return done


# ---- the Runge-Kutta wrappers under compute_scale=None (reference diffusers.py:746-870) -----------------------------------------------
def _backward(model, sample, result, delta: DeltaPoint, noise=None, eta: float = 0):
    "DiffusionModel.backward (reference models.py:68-82): (result - sample * gamma [- noise * zeta]) / delta"
    gamma, dlt = model.gamma(delta, eta), model.delta(delta, eta)
    if noise is not None:
        zeta = model.zeta(delta, eta)
        if zeta != 0:
            return (result - sample * gamma - noise * zeta) / dlt
    return (result - sample * gamma) / dlt


def try_expr(fn, *tensors):
    """a model-transform expression (`to_x`, `from_x`, `forward`, `backward`, `ModelConvert.output_to` called directly) over tensors of one
    16-bit dtype, in the reference's arithmetic -- one launch -- or None (the caller evaluates the fused form)"""
    if not tensors or not all(isinstance(t, torch.Tensor) for t in tensors) or not _eligible(tensors[0], tensors[-1]):
        return None
    first = tensors[0]
    if any(t.dtype != first.dtype or t.shape != first.shape or t.device != first.device or not t.is_contiguous() for t in tensors):
        return None
    try:
        return _express(first, fn, *tensors)
    except _Refused:
        return None


def _express(like: torch.Tensor, fn, *tensors):
    "record fn over the tensors' leaves and run it: one launch (device) / the torch ops themselves (host); a result that IS a leaf comes back as that tensor"
    tape = Tape(like.dtype, like.shape, like.device, require_device=False)
    out = fn(*[tape.leaf(t) for t in tensors])
    (done,) = _execute(tape, [out]) if tape.ops[out.n][0] != _hip.TAPE_LOAD else (tape.leaves[tape.ops[out.n][1]],)
    return done


def _tensor_expression(like: torch.Tensor):
    """`run(fn, *operands)`: the expression as ONE recorded launch while every operand is a contiguous tensor of `like`'s dtype / shape / device and the
    tape takes it; otherwise -- and from then on -- through the operands' own operators, which is the reference's arithmetic op by op"""
    plain = [False]

    def run(fn, *operands):
        if not plain[0] and all(isinstance(t, torch.Tensor) and t.dtype == like.dtype and t.shape == like.shape and t.device == like.device and t.is_contiguous() for t in operands):
            try:
                return _express(like, fn, *operands)
            except _Refused:
                pass
        plain[0] = True
        return fn(*operands)

    return run


def rk_step(wrapper, model_output: torch.Tensor, sample: torch.Tensor, generator):
    """`RKWrapperCore.step` for tensors of one 16-bit dtype under compute_scale=None: the reference's own sequence -- negation, derivative
    conversion, `forward(sample, sumprod(derivatives, row) / fsum(row))` per stage, `backward` for the stages on the clean end -- each as one
    recorded expression (one launch).  Returns (stage input or step result, pred_original_sample), or None when the call is outside this
    mode (the caller then takes the fused path); state (`_derivatives`, `_sample`, `_index`) advances exactly as in the reference."""
    if wrapper.compute_scale is not None or not _eligible(sample, model_output) or not (sample.is_contiguous() and model_output.is_contiguous()):
        return None
    held = list(wrapper._derivatives)
    if any(not isinstance(d, torch.Tensor) or d.dtype != sample.dtype for d in held) or (wrapper._sample is not None and not isinstance(wrapper._sample, torch.Tensor)):
        return None
    nodes, weights = wrapper.tableau()
    if len(weights) + 3 > _hip.TAPE_MAX_INPUTS or 2 * len(weights) + 12 > _hip.TAPE_MAX_OPS:
        return None  # (the 25-35-stage tableaux: more leaves than a tape takes)
    points = [*wrapper.all_points, Point(0, 0, 1)]
    eta = wrapper.stochasticity
    run = _tensor_expression(sample)

    def inside_out(output, space, s0: Point, s1: Point, sn: Point):
        "step_tableau_inside_out (diffusers.py:746-796)"
        wrapper._derivatives.append(output)
        if wrapper._sample is None:
            wrapper._sample = sample
        base, ds = wrapper._sample, list(wrapper._derivatives)
        if len(ds) == len(weights):
            noise = None
            if abs(eta) > 1e-8:
                noise = wrapper.get_step_noise(common.Step.from_int(wrapper._index // wrapper.order, wrapper._steps), base, wrapper.noise_type, wrapper.noise_props, generator, None)
            wrapper._last_noise = noise
            if noise is not None:
                final = run(lambda b, n, *d: _forward(space, b, _sumprod(d, weights), DeltaPoint(s0, s1), n, eta), base, noise, *ds)
            else:
                final = run(lambda b, *d: _forward(space, b, _sumprod(d, weights), DeltaPoint(s0, s1), None, eta), base, *ds)
            wrapper._derivatives = []
            wrapper._sample = None
            return final
        row = nodes[len(ds)][1]
        if not row:
            raise ValueError
        return run(lambda b, *d: _forward(space, b, _sumprod(d, row) / math.fsum(row), DeltaPoint(s0, sn), None, 0), base, *ds)

    output = model_output
    if wrapper.invert_prediction:
        output = run(lambda o: -o, output)
    space = wrapper.model
    if wrapper.derivative_transform:
        at = points[wrapper._index]
        output = run(lambda s_, o: _output_to(wrapper.model, wrapper.derivative_transform, s_, o, at), sample, output)
        space = wrapper.derivative_transform
    n_held = len(held)
    i0, i1, sn = wrapper._index - n_held, wrapper._index + wrapper.order - n_held, wrapper._index + 1
    sampled = inside_out(output, space, points[i0], points[i1], points[sn])
    wrapper._index += 1
    clean = wrapper.schedule.point_0
    while wrapper._index < len(wrapper.all_points) and (
        abs(wrapper.all_points[wrapper._index].timestep - clean.timestep) < 1e-8 or abs(wrapper.all_points[wrapper._index].sigma - clean.sigma) < 1e-8
    ):
        base = sample if wrapper._sample is None else wrapper._sample
        delta = DeltaPoint(points[i0], points[i1])
        synth = run(lambda b, r: _backward(space, b, r, delta), base, sampled)
        sampled = inside_out(synth, space, points[i0], points[i1], points[sn + 1])
        wrapper._index += 1
    return sampled, output


def step_tableau(tableau, sample, model, model_transform, schedule, step, derivative_transform, noise, eta: float, epsilon: float):
    """`functional.step_tableau` (reference functional.py:55-108) for a tensor sample of one 16-bit dtype outside a compute scale: every stage input,
    clean-end derivative and weighted result is one recorded expression in the reference's own order (one launch each); a network output that is not a
    tensor of that dtype sends the rest of the step through plain tensor operators -- the reference's own arithmetic, op by op.  None: not this mode."""
    if not isinstance(sample, torch.Tensor) or not _eligible(sample, sample) or not sample.is_contiguous():
        return None
    if noise is not None and not (isinstance(noise, torch.Tensor) and noise.dtype == sample.dtype and noise.shape == sample.shape and noise.device == sample.device):
        return None
    nodes, weight_rows = tableau[0], tableau[1:]
    if max(len(row) for row in weight_rows) + 3 > _hip.TAPE_MAX_INPUTS or 2 * max(len(row) for row in weight_rows) + 12 > _hip.TAPE_MAX_OPS:
        return None
    if derivative_transform:
        model = models.ModelConvert(model_transform, derivative_transform).wrap_model_call(model)  # (output_to: the tape as well)
        model_transform = derivative_transform
    t0, t1 = step
    s0, s1, *fractions = schedule.ipoints([t0, t1, *(t0 + c * (t1 - t0) for c, _ in nodes)])
    delta = DeltaPoint(s0, s1)
    run = _tensor_expression(sample)

    derivatives: list = []
    for frac, (_c, row) in zip(fractions, nodes):
        if row:
            stage_in = run(lambda b, *d: _forward(model_transform, b, _sumprod(d, row) / math.fsum(row), DeltaPoint(delta.point_from, frac), None, 0), sample, *derivatives)
        else:
            stage_in = sample
        if abs(frac.timestep) < epsilon or abs(frac.sigma) < epsilon:  # never call the network at timestep = 0 or sigma = 0
            derivatives.append(run(lambda b, x: _backward(model_transform, b, x, delta), sample, stage_in))
        else:
            derivatives.append(model(stage_in, *frac))
    if noise is not None:
        return tuple(run(lambda b, n, *d, w=w: _forward(model_transform, b, _sumprod(d, w), delta, n, eta), sample, noise, *derivatives) for w in weight_rows)
    return tuple(run(lambda b, *d, w=w: _forward(model_transform, b, _sumprod(d, w), delta, None, eta), sample, *derivatives) for w in weight_rows)
