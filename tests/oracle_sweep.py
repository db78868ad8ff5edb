"""The sweep fixtures' constructor texts (tests/sweep_grammar.py: W = wrappers, T = samplers, S = schedules, M = models) evaluated into ORACLE objects:
a namespace of small shims that turn each constructor call into the oracle's own description (skr_oracle dict configs / Sched objects / drivers), so that
the oracle replays the reference-recorded sweeps like every other fixture.  (driver() still returns None for a wrapper class without a shim.)"""

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


def _dyn(schedule: _Pending, sampler_order=2, stochasticity=0, model="eps", invert_prediction=False, compute_scale=torch.float32, derivative_transform="data"):
    # (the DynasauRK wrapper reads its stage points off the PRISTINE schedule: diffusers.py:1029-1042 runs functional_interface()'s)
    return lambda steps: OW.RKDriver(lambda st: OK.dynasaur_tableau(st, sampler_order), schedule.build(None), model, derivative_transform, stochasticity, compute=compute_scale, invert=invert_prediction)


W = types.SimpleNamespace(SkrampleWrapperScheduler=_wrapper, RKUltraWrapperScheduler=_rk, DynasauRKWrapperScheduler=_dyn)
NAMES = {"W": W, "T": T, "S": S, "M": M, "torch": torch}


def driver(text: str, steps: int):
    "the oracle's driver of this configuration with its timesteps set, or None (a wrapper the oracle does not model)"
    made = eval(text, NAMES)
    if made is None:
        return None
    d = made(steps)
    d.set_timesteps(steps)
    return d


# ---- tests/golden/native_api16.npz: functional loops and direct calls as oracle calls -----------------------------------------------------------
def _pt(t, s, a):
    from skr_oracle.scalars import Pt

    return Pt(t, s, a)


class _Model:
    "M.<Model>() in a `call` text: the methods forward to the oracle's predictor functions"

    def __init__(self, pred):
        self.pred = pred

    def to_x(self, s, o, p):
        from skr_oracle import predictors as P

        return P.to_x(self.pred, s, o, p)

    def from_x(self, s, x, p):
        from skr_oracle import predictors as P

        return P.from_x(self.pred, s, x, p)

    def forward(self, s, o, d, n=None, eta=0):
        from skr_oracle import predictors as P

        return P.forward(self.pred, s, o, d[0], d[1], n, eta)

    def backward(self, s, r, d, n=None, eta=0):
        from skr_oracle import predictors as P

        return P.backward(self.pred, s, r, d[0], d[1], n, eta)


class _Convert:
    def __init__(self, a: _Model, b: _Model):
        self.a, self.b = a, b

    def output_to(self, s, o, p):
        from skr_oracle import predictors as P

        return P.convert(self.a.pred, self.b.pred, s, o, p, identical=self.a is self.b)

    def output_from(self, s, o, p):
        from skr_oracle import predictors as P

        return P.convert(self.b.pred, self.a.pred, s, o, p, identical=self.a is self.b)


from skr_oracle.scalars import Pt as _Pt  # noqa: E402


class _PointCalls(_Pt):
    def add_noise(self, s, n):
        from skr_oracle.scalars import pt_add_noise

        return pt_add_noise(self, s, n)

    def remove_noise(self, s, n):
        from skr_oracle.scalars import pt_remove_noise

        return pt_remove_noise(self, s, n)


CALL_NAMES = {
    "M": types.SimpleNamespace(
        DataModel=lambda: _Model("data"), NoiseModel=lambda: _Model("eps"), FlowModel=lambda: _Model("flow"), VelocityModel=lambda: _Model("v"), ScaleX=lambda bias=3: _Model(("scalex", bias)),
        ModelConvert=_Convert,
    ),  # fmt: skip
    "Point": lambda t, s, a: _PointCalls(t, s, a),
    "DeltaPoint": lambda a, b: (a, b),
}


def loop(sampler_text: str, model_text: str, schedule_text: str, steps: int, x, net, rng):
    "a functional sampler loop of the fixture through the oracle's loops (rk_loop / adapter_loop)"
    pred = eval(model_text, {"M": M})
    sched = eval(schedule_text, {"S": S}).build(None)  # (no wrapper here: a Karras / Exponential ramp keeps its constructor's `steps`)

    def rk(order=2, stochasticity=0, derivative_transform="data", **kw):
        tab = OK.pick_tableau(order)
        return lambda: OK.rk_loop(lambda st: tab, x, net, pred, sched, steps, rng=rng, deriv=derivative_transform, eta=stochasticity)

    def dyn(order=2, stochasticity=0, derivative_transform="data", **kw):
        return lambda: OK.rk_loop(lambda st: OK.dynasaur_tableau(st, order, **kw), x, net, pred, sched, steps, rng=rng, deriv=derivative_transform, eta=stochasticity)

    run = eval(sampler_text, {"F": types.SimpleNamespace(RKUltra=rk, DynasauRK=dyn), "I": types.SimpleNamespace(StructuredFunctionalAdapter=lambda cfg: lambda: OA.adapter_loop(cfg, x, net, pred, sched, steps, rng=rng)), "T": T, "M": M})
    return run()
