"""The sweep fixtures' constructor texts (tests/sweep_grammar.py: W = wrappers, T = samplers, S = schedules, M = models) evaluated into ORACLE objects:
a namespace of small shims that turn each constructor call into the oracle's own description (skr_oracle dict configs / Sched objects / drivers), so that
the oracle replays the reference-recorded sweeps like every other fixture.  Returns None for what the oracle does not drive (DynasauRK's per-step tableau)."""

from __future__ import annotations

import types

import torch

from skr_oracle import rk as OK
from skr_oracle import samplers as OA
from skr_oracle import schedules as OS
from skr_oracle import wrapper as OW


class _Pending:
    "a schedule description that still needs the run length (the wrappers re-target Karras / Exponential to it: diffusers.py:526-533, 728-735)"

    def __init__(self, build, dynamic: bool):
        self.build, self.dynamic = build, dynamic  # build(steps or None) -> Sched


def _base(fn):
    return lambda **kw: _Pending(lambda steps: fn(**kw), False)


def _sub(fn, takes_steps: bool):
    def make(base: _Pending, **kw):
        def build(steps):
            inner = base.build(None)  # (a sub-schedule sits on a base schedule: nothing dynamic below it)
            if takes_steps and steps is not None and "steps" not in kw:
                return fn(inner, steps=steps, **kw)
            return fn(inner, **kw)

        return _Pending(build, takes_steps)

    return make


def _mod(fn):
    return lambda base, **kw: _Pending(lambda steps: fn(base.build(steps), **kw), base.dynamic)


_beta = lambda base, alpha=0.6, beta=0.6: OS.beta(base, alpha=alpha, beta_=beta)  # noqa: E731
S = types.SimpleNamespace(
    Scaled=_base(OS.scaled), ZSNR=_base(OS.zsnr), Linear=_base(OS.linear), Karras=_sub(OS.karras, True), Exponential=_sub(OS.exponential, True), Beta=_sub(_beta, False),
    Probit=_sub(OS.probit, False), FlowShift=_mod(OS.flowshift), Hyper=_mod(OS.hyper), Sinner=_mod(OS.sinner),
)  # fmt: skip
M = types.SimpleNamespace(DataModel=lambda: "data", NoiseModel=lambda: "eps", FlowModel=lambda: "flow", VelocityModel=lambda: "v", ScaleX=lambda bias=3: ("scalex", bias))
_UNSET = object()


def _deriv(kw: dict) -> dict:
    return {"deriv": kw.pop("derivative_transform")} if "derivative_transform" in kw else {}


def _stated(kind):
    def make(order=None, stochasticity=0, **kw):
        extra = _deriv(kw)
        return OA.make(kind, order, eta=stochasticity, **extra, **kw)

    return make


T = types.SimpleNamespace(
    Euler=lambda stochasticity=0, **kw: OA.make("euler", eta=stochasticity, **_deriv(kw), **kw), DPM=_stated("dpm"), Adams=_stated("adams"), UniP=_stated("unip"), UniPC=_stated("unipc"),
    SPC=lambda **kw: OA.make("spc", **_deriv(kw), **kw),
)  # fmt: skip


def _wrapper(sampler, schedule: _Pending, model="eps", invert_prediction=False, compute_scale=torch.float32):
    return lambda steps: OW.StepDriver(sampler, schedule.build(steps), model, compute=compute_scale, invert=invert_prediction)


def _rk(schedule: _Pending, sampler_order=2, stochasticity=0, model="eps", invert_prediction=False, compute_scale=torch.float32, derivative_transform="data"):
    return lambda steps: OW.RKDriver(OK.pick_tableau(sampler_order), schedule.build(steps), model, derivative_transform, stochasticity, compute=compute_scale, invert=invert_prediction)


W = types.SimpleNamespace(SkrampleWrapperScheduler=_wrapper, RKUltraWrapperScheduler=_rk, DynasauRKWrapperScheduler=lambda *a, **k: None)
NAMES = {"W": W, "T": T, "S": S, "M": M, "torch": torch}


def driver(text: str, steps: int):
    "the oracle's driver of this configuration with its timesteps set, or None (a wrapper the oracle does not model)"
    made = eval(text, NAMES)
    if made is None:
        return None
    d = made(steps)
    d.set_timesteps(steps)
    return d
