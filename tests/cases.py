"""Shared construction tables: the same named configurations for the oracle (skr_oracle) and the
product (skrample_amd), so every parity test builds both sides from one description."""

from __future__ import annotations

import math

import torch

import skrample_amd.scheduling as PS
from skr_oracle import samplers as OA
from skr_oracle import schedules as OS
from skrample_amd.sampling import models as PM
from skrample_amd.sampling import structured as PT

SCHEDULES = {
    # name: (oracle factory, product factory)
    "scaled": (lambda: OS.scaled(), lambda: PS.Scaled()),
    "karras_scaled": (lambda: OS.karras(OS.scaled()), lambda: PS.Karras(PS.Scaled())),
    "linear": (lambda: OS.linear(), lambda: PS.Linear()),
    "zsnr": (lambda: OS.zsnr(), lambda: PS.ZSNR()),
    "flowshift_linear": (lambda: OS.flowshift(OS.linear()), lambda: PS.FlowShift(PS.Linear())),
    "beta_zsnr_flowshift": (lambda: OS.flowshift(OS.beta(OS.zsnr())), lambda: PS.FlowShift(PS.Beta(PS.ZSNR()))),
    "hyper_scaled": (lambda: OS.hyper(OS.scaled()), lambda: PS.Hyper(PS.Scaled())),
    "exponential_scaled": (lambda: OS.exponential(OS.scaled()), lambda: PS.Exponential(PS.Scaled())),
    "sinner_linear": (lambda: OS.sinner(OS.linear()), lambda: PS.Sinner(PS.Linear())),
    "probit_linear": (lambda: OS.probit(OS.linear()), lambda: PS.Probit(PS.Linear())),
    "linear_vp14": (lambda: OS.linear(sigma_start=14.6), lambda: PS.Linear(sigma_start=14.6)),
    "scaled_neg_b1": (lambda: OS.scaled(base_timesteps=-1000, beta_scale=1), lambda: PS.Scaled(base_timesteps=-1000, beta_scale=1)),
}


def oracle_schedule(name: str, steps: int):
    "the wrapper replaces Karras/Exponential.steps by the run length (allow_dynamic); the oracle is told explicitly"
    if name == "karras_scaled":
        return OS.karras(OS.scaled(), steps=steps)
    if name == "exponential_scaled":
        return OS.exponential(OS.scaled(), steps=steps)
    return SCHEDULES[name][0]()


MODELS = {
    "data": ("data", PM.DataModel()),
    "eps": ("eps", PM.NoiseModel()),
    "flow": ("flow", PM.FlowModel()),
    "v": ("v", PM.VelocityModel()),
    "scalex": (("scalex", 3), PM.ScaleX()),
}

SAMPLERS = {
    "euler": (lambda: OA.make("euler"), lambda: PT.Euler()),
    "euler_sde": (lambda: OA.make("euler", eta=1), lambda: PT.Euler(stochasticity=1)),
    "dpm1": (lambda: OA.make("dpm", 1), lambda: PT.DPM(order=1)),
    "dpm1_sde": (lambda: OA.make("dpm", 1, eta=0.5), lambda: PT.DPM(order=1, stochasticity=0.5)),
    "dpm2": (lambda: OA.make("dpm", 2), lambda: PT.DPM(order=2)),
    "dpm2_sde": (lambda: OA.make("dpm", 2, eta=1), lambda: PT.DPM(order=2, stochasticity=1)),
    "dpm3": (lambda: OA.make("dpm", 3), lambda: PT.DPM(order=3)),
    "dpm3_sde": (lambda: OA.make("dpm", 3, eta=0.5), lambda: PT.DPM(order=3, stochasticity=0.5)),
    "adams4": (lambda: OA.make("adams", 4), lambda: PT.Adams(order=4)),
    "adams4_sde": (lambda: OA.make("adams", 4, eta=-1.5), lambda: PT.Adams(order=4, stochasticity=-1.5)),
    "adams9": (lambda: OA.make("adams", 9), lambda: PT.Adams(order=9)),
    "unip2_fast": (lambda: OA.make("unip", 2, fast_solve=True), lambda: PT.UniP(order=2, fast_solve=True)),
    "unip3": (lambda: OA.make("unip", 3), lambda: PT.UniP(order=3)),
    "unip4_sde": (lambda: OA.make("unip", 4, eta=-1.5), lambda: PT.UniP(order=4, stochasticity=-1.5)),
    "unip9": (lambda: OA.make("unip", 9), lambda: PT.UniP(order=9)),
    "unipc1": (lambda: OA.make("unipc", 1), lambda: PT.UniPC(order=1)),
    "unipc1_fast": (lambda: OA.make("unipc", 1, fast_solve=True), lambda: PT.UniPC(order=1, fast_solve=True)),
    "unipc2_fast": (lambda: OA.make("unipc", 2, fast_solve=True), lambda: PT.UniPC(order=2, fast_solve=True)),
    "unipc3": (lambda: OA.make("unipc", 3), lambda: PT.UniPC(order=3)),
    "unipc3_sde": (lambda: OA.make("unipc", 3, eta=1), lambda: PT.UniPC(order=3, stochasticity=1)),
    "unipc9": (lambda: OA.make("unipc", 9), lambda: PT.UniPC(order=9)),
    "unipc2_adams3": (lambda: OA.make("unipc", 2, predictor=OA.make("adams", 3)), lambda: PT.UniPC(order=2, predictor=PT.Adams(order=3))),
    "unipc3_adams2": (lambda: OA.make("unipc", 3, predictor=OA.make("adams", 2)), lambda: PT.UniPC(order=3, predictor=PT.Adams(order=2))),
    "dpm2_deriv_v": (lambda: OA.make("dpm", 2, deriv="v"), lambda: PT.DPM(order=2, derivative_transform=PM.VelocityModel())),
    "unipc3_deriv_flow_sde": (lambda: OA.make("unipc", 3, deriv="flow", eta=0.3), lambda: PT.UniPC(order=3, derivative_transform=PM.FlowModel(), stochasticity=0.3)),
    "adams3_noderiv": (lambda: OA.make("adams", 3, deriv=None), lambda: PT.Adams(order=3, derivative_transform=None)),
    "unipc3_noderiv": (lambda: OA.make("unipc", 3, deriv=None), lambda: PT.UniPC(order=3, derivative_transform=None)),
    "spc": (lambda: OA.make("spc"), lambda: PT.SPC()),
    "spc_bias": (lambda: OA.make("spc", bias=0.3), lambda: PT.SPC(bias=0.3)),
    "spc_power2": (lambda: OA.make("spc", power=2), lambda: PT.SPC(power=2)),
    "spc_power_half_dpm": (lambda: OA.make("spc", power=0.5, predictor=OA.make("dpm", 2, eta=0.5), corrector=OA.make("adams", 2)), lambda: PT.SPC(power=0.5, predictor=PT.DPM(order=2, stochasticity=0.5), corrector=PT.Adams(order=2))),
    "spc_dpm_unip": (
        lambda: OA.make("spc", predictor=OA.make("dpm", 2, eta=1), corrector=OA.make("unip", 3), adaptive=False, invert=True),
        lambda: PT.SPC(predictor=PT.DPM(order=2, stochasticity=1), corrector=PT.UniP(order=3), adaptive=False, invert=True),
    ),
}


def fake_model(x, t, s, a):
    return x - math.sin(t)


def from_bits(arr, dtype: torch.dtype) -> torch.Tensor:
    "inverse of tools/make_golden.py::bits"
    t = torch.from_numpy(arr.copy())
    if dtype == torch.bfloat16:
        return t.view(torch.bfloat16)
    if dtype == torch.float16:
        return t.view(torch.float16) if t.dtype == torch.int16 else t
    return t


def bf16_ulp(ref: torch.Tensor) -> torch.Tensor:
    "one bf16 unit in the last place at |ref|"
    mag = ref.float().abs().clamp_min(2.0**-126)
    return torch.exp2(torch.floor(torch.log2(mag)) - 7)


# ---- tests/golden/native16.npz: the sampler-level API called directly on 16-bit tensors (reference-recorded) ----------------
NATIVE16_ORACLE = {
    "euler": lambda OA: OA.make("euler"),
    "euler_sde_v": lambda OA: OA.make("euler", eta=1),
    "dpm2_sde": lambda OA: OA.make("dpm", 2, eta=1),
    "dpm3_flow": lambda OA: OA.make("dpm", 3),
    "adams3": lambda OA: OA.make("adams", 3),
    "unipc3_flow": lambda OA: OA.make("unipc", 3),
}
NATIVE16_TAGS = [f"{n}/{d}" for n in NATIVE16_ORACLE for d in ("bf16", "f16")]


def native16_case(blob, tag):
    "(dtype, steps, model name, schedule name, sampler constructor text, per-step tensors as 16-bit torch tensors)"
    dt = torch.bfloat16 if tag.endswith("/bf16") else torch.float16
    expr, mname, sname, steps = (str(v) for v in blob[tag + "/meta"])
    view = lambda key: [torch.from_numpy(a.copy()).view(dt) for a in blob[f"{tag}/{key}"]]  # noqa: E731
    return dt, int(steps), mname, sname, expr, {k: view(k) for k in ("x", "out", "noise", "final", "prediction")}
