"""Step programs: a solver step lowered once, replayed by pointer binding.

The first time the scheduler wrapper meets a (step index, history layout, dtypes, shape) it runs the
sampler's lazy algebra normally and records the single fused launch it produced (`_hip.trace`).  That
launch -- the filled `skr_step_plan` plus, for every operand, *where it came from* (current sample / model
output / noise, or field of the k-th history record) -- is a `StepProgram`.  Later calls with the same key
skip the Python algebra entirely: resolve the roles to today's tensors, patch the Philox stream ids, launch.
Coefficients depend only on the key (schedule, step index and history steps), never on tensor contents, so
replay is exact.  Anything unusual (operands that needed a copy, extra launches, foreign tensors) simply
leaves the step un-cached and on the normal path.
"""

from __future__ import annotations

import ctypes
from typing import Any, Sequence

import torch

from .. import _hip
from .lazy import LazyTensor, Lin, PhiloxNoise, empty_output
from .structured import SKSamples

Role = tuple  # ("x",) ("o",) ("n",) ("px", k) ("pi", k) ("po", k) ("pn", k)   k = negative index into the history


class Roles:
    "resolves roles against the current call"

    __slots__ = ("sample", "output", "noise", "previous", "raw_outputs", "raw_samples")

    def __init__(self, sample, output, noise, previous: Sequence[SKSamples], raw_outputs: Sequence[torch.Tensor], raw_samples: Sequence[torch.Tensor]):
        self.sample, self.output, self.noise, self.previous, self.raw_outputs, self.raw_samples = sample, output, noise, previous, raw_outputs, raw_samples

    def get(self, role: Role):
        kind = role[0]
        if kind == "x":
            return self.sample
        if kind == "o":
            return self.output
        if kind == "n":
            return self.noise
        k = role[1]
        if kind == "px":
            return self.previous[k].sample
        if kind == "po":
            return self.raw_outputs[k]
        if kind == "pi":
            return self.raw_samples[k]
        return self.previous[k].noise

    def table(self) -> dict[int, Role]:
        "id(object) -> role for everything a launch could legitimately read"
        out: dict[int, Role] = {}
        for k in range(-len(self.previous), 0):
            rec = self.previous[k]
            if isinstance(rec.sample, torch.Tensor):
                out[id(rec.sample)] = ("px", k)
            out[id(self.raw_outputs[k])] = ("po", k)
            out[id(self.raw_samples[k])] = ("pi", k)  # what the model saw (differs from rec.sample for UniPC / SPC)
            if rec.noise is not None:
                out[id(rec.noise)] = ("pn", k)
        if self.noise is not None:
            out[id(self.noise)] = ("n",)
        out[id(self.output)] = ("o",)
        out[id(self.sample)] = ("x",)
        return out


class StepProgram:
    __slots__ = ("plan", "roles", "dtypes", "shape", "numel", "out_dtypes", "noise_roles", "final_out", "state_out", "pred", "ptr_array")

    def __init__(self):
        self.ptr_array = None

    @staticmethod
    def build(trace_entry, roles: Roles, record: SKSamples, prediction_in) -> "StepProgram | None":
        plan, inputs, out0, out1, seeds, numel = trace_entry
        table = roles.table()
        prog = StepProgram()
        prog.roles = []
        for t in inputs:
            role = table.get(id(t))
            if role is None:
                return None  # a temporary (copy of a misaligned view, realised noise, cast) -- not replayable
            prog.roles.append(role)
        prog.dtypes = [t.dtype for t in inputs]
        prog.shape, prog.numel = tuple(record.final.shape), numel
        prog.out_dtypes = (out0.dtype, out1.dtype if out1 is not None else None)
        # which Philox draw feeds which output
        prog.noise_roles = [None, None]
        if plan.noise_mode == 1:
            cands = [(r, roles.get(r)) for r in [("n",)] + [("pn", k) for k in range(-len(roles.previous), 0)]]
            cands = [(r, n) for r, n in cands if isinstance(n, PhiloxNoise)]
            for slot, (zeta, stream) in enumerate(((plan.zeta0, plan.stream0), (plan.zeta1, plan.stream1))):
                if zeta != 0.0:
                    hit = [r for r, n in cands if n.stream == stream and n.seeds is seeds]
                    if len(hit) != 1:
                        return None
                    prog.noise_roles[slot] = hit[0]
        # outputs
        if record.final is out0 and out1 is None:
            prog.final_out, prog.state_out = 0, None
        elif record.final is out1:
            prog.final_out = 1
            prog.state_out = 0 if record.sample is out0 else None
            if prog.state_out is None:
                return None
        else:
            return None
        if prog.state_out is None and record.sample is not roles.sample:
            return None
        # the record's prediction: the caller's object itself, or a form over replayable leaves
        if record.prediction is prediction_in:
            prog.pred = None
        elif isinstance(record.prediction, LazyTensor):
            terms = []
            for leaf, c in record.prediction.form.expanded().terms.values():
                role = table.get(id(leaf))
                if role is None:
                    return None
                terms.append((role, c))
            prog.pred = (terms, record.prediction.dtype)
        else:
            return None
        prog.plan = plan
        return prog

    def run(self, roles: Roles, step, prediction_in, device: torch.device) -> SKSamples | None:
        "replay; returns None (caller falls back to the normal path) if today's operands do not fit"
        ops = []
        for role, dt in zip(self.roles, self.dtypes):
            t = roles.get(role)
            if not isinstance(t, torch.Tensor) or t.dtype != dt or t.numel() != self.numel or not t.is_contiguous() or t.data_ptr() % 16 or t.device != device:
                return None
            ops.append(t)
        plan = self.plan
        seeds_ptr = None
        for slot, role in enumerate(self.noise_roles):
            if role is None:
                continue
            nz = roles.get(role)
            if not isinstance(nz, PhiloxNoise) or not nz.fusable() or nz.shape != self.shape:
                return None
            if slot == 0:
                plan.stream0 = nz.stream
            else:
                plan.stream1 = nz.stream
            ptr = nz.seeds.data_ptr()
            if seeds_ptr is not None and ptr != seeds_ptr:
                return None
            seeds_ptr = ptr
        out0 = empty_output(self.shape, self.out_dtypes[0], device)
        out1 = empty_output(self.shape, self.out_dtypes[1], device) if self.out_dtypes[1] is not None else None
        n = len(ops)
        if self.ptr_array is None:
            self.ptr_array = (ctypes.c_void_p * max(n, 1))()
        arr = self.ptr_array
        for i in range(n):
            arr[i] = ops[i].data_ptr()
        lib = _hip.load()
        if _hip.trace is not None:
            _hip.trace.append((plan, ops, out0, out1, None, self.numel))
        status = _hip.step_launch_raw(plan, arr, out0.data_ptr(), out1.data_ptr() if out1 is not None else None, seeds_ptr, self.numel, _hip.current_stream_ptr(device))
        _hip.check(status, "skr_step_launch")
        outs = (out0, out1)
        final = outs[self.final_out]
        sample = outs[self.state_out] if self.state_out is not None else roles.sample
        prediction = prediction_in if self.pred is None else self.lazy_prediction(roles, self.shape, device)
        return SKSamples(sample, prediction, step, roles.noise, final)

    def lazy_prediction(self, roles: Roles, shape, device: torch.device) -> LazyTensor:
        "the record's prediction as a form over today's operands, materialised only if somebody reads it"
        terms, dtype = self.pred
        bound = [(roles.get(r), c) for r, c in terms]  # bind now: the history window moves on

        def make_form(bound=bound):
            form = None
            for leaf, c in bound:
                piece = Lin.leaf(leaf) * c
                form = piece if form is None else form + piece
            return form

        return LazyTensor(None, dtype, form_fn=make_form, shape=tuple(shape), device=device, leaves=[leaf for leaf, _ in bound], acc_f64=bool(self.plan.acc_f64))
