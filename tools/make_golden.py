#!/usr/bin/env python3
"""Generate tests/golden/* from the reference itself (build container only).

Imports /root/reference through tools/ref_loader.py (in-memory PEP-695 de-sugaring; no reference
source is stored), drives the reference's own classes on seeded inputs and writes ONLY numbers:
inputs, consumed random draws and the reference's outputs.  Re-run:  python tools/make_golden.py

Outputs
  reference_kats.json   numbers the reference's own tests hold (self_sampling.py:57-82,
                        self_scheduling.py:30-45, miscellaneous.py:11), extracted from the test text
  tables.json           wrapper timesteps / sigmas / schedule_np, Gamma/Delta/zeta, effective_order,
                        RK all_points, every built-in tableau
  steps_cfg{1..5}.npz   per-step (x_t, model_out, noise) -> (prev_sample, pred_original_sample)
                        through SkrampleWrapperScheduler.step / RKUltraWrapperScheduler.step
  steps_extra.npz       same for further sampler/model/eta combinations (steps_extra2..4: later rounds' additions)
  noise.npz             Offset / Pyramid / Colored outputs with the raw draws they consumed
"""

from __future__ import annotations

import ast
import dataclasses
import json
import math
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("SKR_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))

import ref_loader  # noqa: E402

ref_loader.install()

import skrample.diffusers as RD  # noqa: E402
import skrample.pytorch.noise as RN  # noqa: E402
import skrample.scheduling as RS  # noqa: E402
from skrample.common import DeltaPoint, Step, bashforth  # noqa: E402
from skrample.sampling import functional, models, structured, tableaux  # noqa: E402

REF = ref_loader.REFERENCE_ROOT


def bits(t: torch.Tensor) -> np.ndarray:
    "lossless numpy view (bf16 -> int16 bit pattern)"
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().copy()
    return t.contiguous().numpy().copy()


# ---------------------------------------------------------------------------------------------------
def kats() -> None:
    out: dict = {}
    text = open(os.path.join(REF, "tests", "self_sampling.py"), encoding="utf-8").read()
    rows = re.findall(r"\((\w+)\.(\w+), scheduling\.(\w+), models\.(\w+)\): (\[[^\]]*\])", text)
    out["sampler_trajectories"] = {f"{s}/{sch}/{m}": ast.literal_eval(v) for _, s, sch, m, v in rows}
    assert len(out["sampler_trajectories"]) == 24
    text = open(os.path.join(REF, "tests", "self_scheduling.py"), encoding="utf-8").read()
    rows = re.findall(r"^    ([A-Za-z()]+\(\)\)+): (\[\[.*\]\]),", text, flags=re.M)
    out["schedule_points"] = {k: ast.literal_eval(v) for k, v in rows}
    assert len(out["schedule_points"]) == 14, len(out["schedule_points"])
    out["bashforth"] = [[1], [3 / 2, -1 / 2], [23 / 12, -4 / 3, 5 / 12], [55 / 24, -59 / 24, 37 / 24, -3 / 8]]  # miscellaneous.py:11
    out["measured_steps"] = 7
    out["measured_seed"] = 42
    json.dump(out, open(os.path.join(OUT, "reference_kats.json"), "w"), indent=0)


# ---------------------------------------------------------------------------------------------------
CFG_SCHEDULES = {
    "scaled": lambda: RS.Scaled(),
    "karras_scaled": lambda: RS.Karras(RS.Scaled()),
    "linear": lambda: RS.Linear(),
    "zsnr": lambda: RS.ZSNR(),
    "flowshift_linear": lambda: RS.FlowShift(RS.Linear()),
    "beta_zsnr_flowshift": lambda: RS.FlowShift(RS.Beta(RS.ZSNR())),
    "hyper_scaled": lambda: RS.Hyper(RS.Scaled()),
    "exponential_scaled": lambda: RS.Exponential(RS.Scaled()),
    "sinner_linear": lambda: RS.Sinner(RS.Linear()),
    "probit_linear": lambda: RS.Probit(RS.Linear()),
    "linear_vp14": lambda: RS.Linear(sigma_start=14.6),
    "scaled_neg_b1": lambda: RS.Scaled(base_timesteps=-1000, beta_scale=1),
}
MODELS = {"data": models.DataModel(), "eps": models.NoiseModel(), "flow": models.FlowModel(), "v": models.VelocityModel(), "scalex": models.ScaleX()}


def tables() -> None:
    out: dict = {"wrapper": {}, "gdz": [], "effective_order": [], "tableaux": {}, "rk_points": {}}
    for name, mk in CFG_SCHEDULES.items():
        for n in (1, 2, 7, 20, 50):
            w = RD.SkrampleWrapperScheduler(structured.Euler(), mk())
            w.set_timesteps(n)
            out["wrapper"][f"{name}/{n}"] = {
                "timesteps": w.timesteps.tolist(),
                "sigmas": w.sigmas.tolist(),
                "schedule_np": w.schedule_np.tolist(),
                "point_0": list(w.schedule.point(0)),
                "point_1": list(w.schedule.point(1)),
                "ipoint_0.37": list(w.schedule.ipoint(0.37)),
            }
    for sname in ("scaled", "linear", "zsnr", "karras_scaled"):
        sch = CFG_SCHEDULES[sname]()
        for mname, m in MODELS.items():
            for eta in (-1.5, 0, 0.5, 1):
                for a, b in ((0.0, 0.05), (0.4, 0.45), (0.9, 1.0), (0.35, 0.7)):
                    dp = DeltaPoint(*sch.ipoints([a, b]))
                    try:
                        row = [m.gamma(dp, eta), m.delta(dp, eta), m.zeta(dp, eta)]
                    except ZeroDivisionError:
                        row = None
                    if row is not None and not all(math.isfinite(v) for v in row):
                        row = [repr(v) for v in row]
                    out["gdz"].append({"schedule": sname, "model": mname, "eta": eta, "step": [a, b], "gdz": row})
    for cls, order in ((structured.DPM, 3), (structured.Adams, 4), (structured.UniP, 9), (structured.UniPC, 3)):
        for steps in (1, 2, 5, 20):
            for nprev in (0, 1, 3, 12):
                s = cls(order=order)
                out["effective_order"].append(
                    {
                        "sampler": cls.__name__,
                        "order": order,
                        "steps": steps,
                        "n_previous": nprev,
                        "eo": [s.effective_order(Step.from_int(i, steps), [None] * nprev) for i in range(steps)],
                        "require_previous": s.require_previous,
                    }
                )
    for group in (tableaux.RK1, tableaux.RK2, tableaux.RK3, tableaux.RK4, tableaux.RKZ, tableaux.RKE2, tableaux.RKE3, tableaux.RKE5, tableaux.SSP, tableaux.WSO, tableaux.Shanks1965):
        for member in group:
            tab = member.tableau()
            out["tableaux"][f"{group.__name__}.{member.name}"] = {
                "c": [float(s.c) for s in tab.stages],
                "a": [[float(v) for v in s.a] for s in tab.stages],
                "b": [float(v) for v in tab.weights],
                **({"e": [float(v) for v in tab.error_weights]} if hasattr(tab, "error_weights") else {}),
            }
    out["default_providers"] = {str(k): f"{type(v).__name__}.{v.name}" for k, v in functional.DEFAULT_PROVIDERS.items()}
    out["stable_providers"] = {str(k): f"{type(v).__name__}.{v.name}" for k, v in functional.STABLE_PROVIDERS.items()}
    for x in (0.1, 0.25, 0.4):
        out["tableaux"][f"gen.rk2({x})"] = _tab(tableaux.providers.rk2_tableau(x))
        out["tableaux"][f"gen.ees25({x})"] = _tab(tableaux.providers.ees25_tableau(x))
        out["tableaux"][f"gen.ees27({x})"] = _tab(tableaux.providers.ees27_tableau(x))
    out["tableaux"]["gen.rk3(0.5,0.75)"] = _tab(tableaux.providers.rk3_tableau(0.5, 0.75))
    out["tableaux"]["gen.rk4(0.4,0.6)"] = _tab(tableaux.providers.rk4_tableau(0.4, 0.6))
    for sname in ("scaled", "linear", "sinner_linear"):
        for order in (1, 2, 3, 4, 5, 6, 99):
            for steps in (1, 3, 7):
                w = RD.RKUltraWrapperScheduler(CFG_SCHEDULES[sname](), sampler_order=order)
                w.set_timesteps(steps)
                out["rk_points"][f"rku/{sname}/{order}/{steps}"] = {"all": [list(p) for p in w.all_points], "timesteps": w.timesteps.tolist(), "order": w.order}
        for order in (2, 3, 4):
            for steps in (1, 3, 7):
                w = RD.DynasauRKWrapperScheduler(CFG_SCHEDULES[sname](), sampler_order=order, model=models.FlowModel() if "linear" in sname else models.NoiseModel())
                w.set_timesteps(steps)
                out["rk_points"][f"dyn/{sname}/{order}/{steps}"] = {"all": [list(p) for p in w.all_points], "timesteps": w.timesteps.tolist(), "order": w.order}
    json.dump(out, open(os.path.join(OUT, "tables.json"), "w"))


def _tab(tab) -> dict:
    return {"c": [float(s.c) for s in tab.stages], "a": [[float(v) for v in s.a] for s in tab.stages], "b": [float(v) for v in tab.weights]}


# ---------------------------------------------------------------------------------------------------
class _Injected:
    "stands in for BatchTensorNoise: hands back pre-drawn noise so the fixture knows what was consumed"

    def __init__(self, draws):
        self.draws = list(draws)

    def generate(self, step):
        return self.draws.pop(0)


def run_wrapper(w, B, unit, steps, dtype, seed, noise_fn=None):
    g = torch.Generator().manual_seed(seed)
    w.set_timesteps(steps)
    n_calls = len(w.timesteps)
    x = torch.randn([B, *unit], generator=g).to(dtype)
    outs = [torch.randn([B, *unit], generator=g).to(dtype) for _ in range(n_calls)]
    noises = [(noise_fn(i, g) if noise_fn else torch.randn([B, *unit], generator=g)) for i in range(n_calls)]
    w._noise_generator = _Injected(noises)
    rec = {"x0": bits(x), "outs": np.stack([bits(o) for o in outs]), "noises": np.stack([bits(n) for n in noises]), "timesteps": w.timesteps.numpy().copy()}
    prevs, preds = [], []
    for t, o in zip(w.timesteps, outs):
        prev, pred = w.step(o, t, x, return_dict=False)
        prevs.append(bits(prev))
        preds.append(bits(pred))
        x = prev
    rec["prev"] = np.stack(prevs)
    rec["pred"] = np.stack(preds)
    rec["noise_used"] = np.asarray(n_calls - len(w._noise_generator.draws))
    if int(rec["noise_used"]) == 0:
        rec["noises"] = np.zeros((0,), dtype=np.float32)  # deterministic sampler: nothing was consumed
    return rec


def steps() -> None:
    bf16, f32 = torch.bfloat16, torch.float32
    cases = {
        # the five BASELINE.json configs at reduced shape
        "cfg1": (lambda: RD.SkrampleWrapperScheduler(structured.Euler(), RS.Scaled()), 1, (4, 64, 64), 5, f32),
        "cfg2": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2, stochasticity=1), RS.Karras(RS.Scaled())), 2, (4, 16, 16), 8, bf16),
        "cfg3": (lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=3, stochasticity=1), RS.Linear(), models.FlowModel()), 2, (16, 16, 16), 8, bf16),
        "cfg4": (lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=4), RS.ZSNR(), models.VelocityModel()), 2, (4, 16, 16), 8, bf16),
        "cfg5": (lambda: RD.RKUltraWrapperScheduler(RS.Scaled(), sampler_order=6, stochasticity=1), 2, (4, 16, 16), 3, bf16),
    }
    for name, (mk, B, unit, n, dt) in cases.items():
        np.savez_compressed(os.path.join(OUT, f"steps_{name}.npz"), **run_wrapper(mk(), B, unit, n, dt, seed=1000 + len(name)))

    extra = {
        "euler_sde_v_zsnr": (lambda: RD.SkrampleWrapperScheduler(structured.Euler(stochasticity=1), RS.ZSNR(), models.VelocityModel()), bf16),
        "dpm1_ode_eps": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=1), RS.Scaled()), bf16),
        "dpm3_sde_eps": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=3, stochasticity=0.5), RS.Scaled()), f32),
        "dpm2_flow_shift": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2), RS.FlowShift(RS.Linear()), models.FlowModel()), bf16),
        "adams9_data": (lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=9), RS.Scaled(), models.DataModel()), f32),
        "unip4_eps": (lambda: RD.SkrampleWrapperScheduler(structured.UniP(order=4, stochasticity=-1.5), RS.Scaled()), f32),
        "unipc2_fast_v": (lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=2, fast_solve=True), RS.Scaled(), models.VelocityModel()), bf16),
        "unipc3_adams_pred": (lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=3, predictor=structured.Adams(order=2)), RS.Linear(), models.FlowModel()), f32),
        "dpm2_deriv_v": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2, derivative_transform=models.VelocityModel()), RS.Scaled()), f32),
        "adams3_noderiv": (lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=3, derivative_transform=None), RS.Scaled()), f32),
        "euler_invert": (lambda: RD.SkrampleWrapperScheduler(structured.Euler(), RS.Scaled(), invert_prediction=True), bf16),
        "dpm2_f16": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2, stochasticity=1), RS.Scaled()), torch.float16),
        "dpm2_f64": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2, stochasticity=1), RS.Scaled(), compute_scale=torch.float64), torch.float64),
        "rku2_ode_flow": (lambda: RD.RKUltraWrapperScheduler(RS.Linear(), sampler_order=2, model=models.FlowModel()), bf16),
        "rku4_sde_v": (lambda: RD.RKUltraWrapperScheduler(RS.Scaled(), sampler_order=4, stochasticity=0.5, model=models.VelocityModel()), f32),
        "rku5_noderiv": (lambda: RD.RKUltraWrapperScheduler(RS.Scaled(), sampler_order=5, derivative_transform=None), f32),
    }
    blob = {}
    for i, (name, (mk, dt)) in enumerate(extra.items()):
        rec = run_wrapper(mk(), 2, (4, 8, 8), 7 if "rku" not in name else 3, dt, seed=2000 + i)
        if dt == torch.float16:
            rec = {k: (v.view(np.int16) if v.dtype == np.float16 else v) for k, v in rec.items()}
        for k, v in rec.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "steps_extra.npz"), **blob)

    # round 2: the predictor-corrector blend (SPC), DynasauRK, nested predictors, the remaining sub-schedules / modifiers
    extra2 = {
        "spc_default": (lambda: RD.SkrampleWrapperScheduler(structured.SPC(), RS.Scaled()), bf16),
        "spc_power2_bias": (lambda: RD.SkrampleWrapperScheduler(structured.SPC(power=2, bias=0.3, adaptive=False), RS.Scaled()), f32),
        "spc_invert_dpm_unip": (lambda: RD.SkrampleWrapperScheduler(structured.SPC(predictor=structured.DPM(order=2), corrector=structured.UniP(order=3), invert=True), RS.Karras(RS.Scaled())), f32),
        "spc_sde_v": (lambda: RD.SkrampleWrapperScheduler(structured.SPC(predictor=structured.Euler(stochasticity=1)), RS.ZSNR(), models.VelocityModel()), bf16),
        "spc_flow_power_half": (lambda: RD.SkrampleWrapperScheduler(structured.SPC(power=0.5, bias=-0.2), RS.Linear(), models.FlowModel()), f32),
        "unipc3_dpm_pred_sde": (lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=3, stochasticity=1, predictor=structured.DPM(order=2, stochasticity=1)), RS.Karras(RS.Scaled())), bf16),
        "dpm2_exponential": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2), RS.Exponential(RS.Scaled())), f32),
        "adams2_beta_zsnr": (lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=2), RS.Beta(RS.ZSNR()), models.VelocityModel()), f32),
        "euler_probit_flow": (lambda: RD.SkrampleWrapperScheduler(structured.Euler(), RS.Probit(RS.Linear()), models.FlowModel()), bf16),
        "dpm2_hyper": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2, stochasticity=0.3), RS.Hyper(RS.Scaled())), f32),
        "unip2_sinner_flow": (lambda: RD.SkrampleWrapperScheduler(structured.UniP(order=2), RS.Sinner(RS.Linear()), models.FlowModel()), f32),
        "dyn3_flow": (lambda: RD.DynasauRKWrapperScheduler(RS.Linear(), sampler_order=3, model=models.FlowModel()), f32),
        "dyn2_sde_eps": (lambda: RD.DynasauRKWrapperScheduler(RS.Scaled(), sampler_order=2, stochasticity=0.5), bf16),
        "dyn4_v": (lambda: RD.DynasauRKWrapperScheduler(RS.Scaled(), sampler_order=4, model=models.VelocityModel()), f32),
    }
    blob = {}
    for i, (name, (mk, dt)) in enumerate(extra2.items()):
        rec = run_wrapper(mk(), 2, (4, 8, 8), 7 if "dyn" not in name else 3, dt, seed=3000 + i)
        for k, v in rec.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "steps_extra2.npz"), **blob)

    # round 3: the high orders north_star names (Adams-Bashforth up to 9, UniPC / UniP beyond 3), on ONE 2048-element sample so
    # that the replay takes the compile-time one-trip kernels (whole 2048-element chunks) -- 10 to 22 operands per launch
    extra3 = {
        "unipc6_sde_eps": (lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=6, stochasticity=1), RS.Scaled()), bf16),
        "adams9_eps_karras": (lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=9), RS.Karras(RS.Scaled())), bf16),
        "adams6_v_zsnr": (lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=6), RS.ZSNR(), models.VelocityModel()), bf16),
        "unip7_flow": (lambda: RD.SkrampleWrapperScheduler(structured.UniP(order=7), RS.Linear(), models.FlowModel()), f32),
        "unipc9_flow": (lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=9), RS.Linear(), models.FlowModel()), bf16),
    }
    blob = {}
    for i, (name, (mk, dt)) in enumerate(extra3.items()):
        rec = run_wrapper(mk(), 1, (4, 16, 32), 12, dt, seed=4000 + i)
        for k, v in rec.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "steps_extra3.npz"), **blob)

    # round 5: compute_scale=float64 over 16-bit latents (diffusers.py:575-599 casts whatever it is handed to compute_scale and the
    # results back to the model output's dtype)
    f16, f64 = torch.float16, torch.float64
    extra4 = {
        "dpm2_f64_bf16": (lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2, stochasticity=1), RS.Scaled(), compute_scale=f64), bf16),
        "unipc3_f64_f16": (lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=3, stochasticity=1), RS.Linear(), models.FlowModel(), compute_scale=f64), f16),
        "adams4_f64_bf16": (lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=4), RS.ZSNR(), models.VelocityModel(), compute_scale=f64), bf16),
        "rku4_f64_bf16": (lambda: RD.RKUltraWrapperScheduler(RS.Scaled(), sampler_order=4, stochasticity=0.5, compute_scale=f64), bf16),
    }
    blob = {}
    for i, (name, (mk, dt)) in enumerate(extra4.items()):
        rec = run_wrapper(mk(), 2, (4, 8, 8), 7 if "rku" not in name else 3, dt, seed=6000 + i)
        if dt == torch.float16:
            rec = {k: (v.view(np.int16) if v.dtype == np.float16 else v) for k, v in rec.items()}
        for k, v in rec.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "steps_extra4.npz"), **blob)


# ---------------------------------------------------------------------------------------------------
# round 5: a seeded random sweep over the wrapper's whole configuration space (sampler x nesting x schedule x modifier x predictor x
# eta x dtype x compute_scale x ragged shape x run length), each case stepped through the imported reference.  A case is stored as
# the TEXT of its constructor in a neutral vocabulary (W = diffusers module, T = structured samplers, S = schedules, M = models), so
# the test builds the product's object from the same words.
from sweep_grammar import native_spec as _native_spec  # noqa: E402
from sweep_grammar import sweep_spec as _sweep_spec  # noqa: E402  (tests/sweep_grammar.py: shared with the GPU box's device-vs-host soak)


def sweep(count: int = 64) -> None:
    import random

    names = {"W": RD, "T": structured, "S": RS, "M": models, "torch": torch}
    blob, meta, refused, seed = {}, [], [], 0
    while len(meta) < count:
        seed += 1
        text, dtype, shape, steps_n = _sweep_spec(random.Random(7000 + seed))
        dt = getattr(torch, dtype)
        try:
            rec = run_wrapper(eval(text, names), shape[0], shape[1:], steps_n, dt, seed=7000 + seed)
        except (ZeroDivisionError, ValueError, AssertionError, IndexError, AttributeError, TypeError) as err:
            refused.append({"text": text, "dtype": dtype, "shape": list(shape), "steps": steps_n, "error": type(err).__name__})
            continue
        view = lambda a: torch.from_numpy(a).view(dt) if dt in (torch.bfloat16, torch.float16) and a.dtype != np.float16 else torch.from_numpy(np.asarray(a))  # noqa: E731
        if not all(torch.isfinite(view(rec[k]).float()).all() for k in ("prev", "pred")):
            continue  # (singular points, e.g. alpha = 0 at the first step: INTEGRATION.md "differences a caller can observe")
        if dt == torch.float16:
            rec = {k: (v.view(np.int16) if v.dtype == np.float16 else v) for k, v in rec.items()}
        tag = f"s{len(meta):02d}"
        meta.append({"tag": tag, "text": text, "dtype": dtype, "shape": list(shape), "steps": steps_n, "seed": 7000 + seed})
        for k, v in rec.items():
            blob[f"{tag}/{k}"] = v
    blob["meta"] = np.asarray(json.dumps(meta))
    blob["refused"] = np.asarray(json.dumps(refused))  # configurations the reference itself refuses (an exception out of set_timesteps / step)
    np.savez_compressed(os.path.join(OUT, "steps_sweep.npz"), **blob)


def sweep_native(count: int = 32) -> None:
    "the sweep's grammar under compute_scale=None on 16-bit tensors (noise handed over in the tensor dtype, as get_step_noise does: diffusers.py:346)"
    import random

    names = {"W": RD, "T": structured, "S": RS, "M": models, "torch": torch}
    blob, meta, seed = {}, [], 0
    while len(meta) < count:
        seed += 1
        text, dtype, shape, steps_n = _native_spec(random.Random(8000 + seed))
        dt = getattr(torch, dtype)
        try:
            rec = run_wrapper(eval(text, names), shape[0], shape[1:], steps_n, dt, seed=8000 + seed, noise_fn=lambda i, g: torch.randn(shape, generator=g).to(dt))
        except (ZeroDivisionError, ValueError, AssertionError, IndexError, AttributeError, TypeError):
            continue
        as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).view(dt)  # noqa: E731
        if not all(torch.isfinite(as_t(rec[k]).float()).all() for k in ("prev", "pred")):
            continue
        rec = {k: (v.view(np.int16) if v.dtype == np.float16 else v) for k, v in rec.items()}
        tag = f"n{len(meta):02d}"
        meta.append({"tag": tag, "text": text, "dtype": dtype, "shape": list(shape), "steps": steps_n, "seed": 8000 + seed})
        for k, v in rec.items():
            blob[f"{tag}/{k}"] = v
    blob["meta"] = np.asarray(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "steps_sweep_native.npz"), **blob)


def native_api16() -> None:
    """The reference's generic tensor arithmetic on 16-bit tensors beyond the samplers: the functional samplers' loops (RKUltra, DynasauRK, the structured
    adapter) over a network both devices compute alike, the model transforms called directly, Point.add_noise / remove_noise.  Every call stored with the text
    that makes it (F = functional module, I = interface, T = samplers, S = schedules, M = models; s / o / n = the three recorded operands)."""
    from skrample.common import DeltaPoint as DP
    from skrample.common import Point as PT_
    from skrample.sampling import interface

    names = {"F": functional, "I": interface, "T": structured, "S": RS, "M": models, "Point": PT_, "DeltaPoint": DP}
    loops = [
        ("F.RKUltra(order=4)", "M.NoiseModel()", "S.Scaled()", 5),
        ("F.RKUltra(order=3, stochasticity=1, derivative_transform=M.VelocityModel())", "M.FlowModel()", "S.Linear()", 4),
        ("F.RKUltra(order=6, stochasticity=0.5)", "M.VelocityModel()", "S.ZSNR()", 3),
        ("F.DynasauRK(order=3)", "M.VelocityModel()", "S.Karras(S.Scaled())", 5),
        ("F.DynasauRK(order=2, stochasticity=0.3, invert=True)", "M.NoiseModel()", "S.Scaled()", 4),
        ("I.StructuredFunctionalAdapter(T.UniPC(order=3, stochasticity=0.5))", "M.NoiseModel()", "S.Scaled()", 6),
        ("I.StructuredFunctionalAdapter(T.DPM(order=2, stochasticity=1))", "M.FlowModel()", "S.FlowShift(S.Linear())", 5),
    ]
    calls = [
        "M.NoiseModel().to_x(s, o, Point(500.0, 0.6, 0.8))", "M.FlowModel().from_x(s, o, Point(500.0, 0.6, 0.4))", "M.VelocityModel().to_x(s, o, Point(500.0, 0.6, 0.8))",
        "M.ScaleX(bias=-1.5).from_x(s, o, Point(500.0, 0.6, 0.8))", "M.NoiseModel().forward(s, o, DeltaPoint(Point(500.0, 0.6, 0.8), Point(300.0, 0.3, 0.9539392014169456)), n, 0.5)",
        "M.FlowModel().forward(s, o, DeltaPoint(Point(500.0, 0.6, 0.4), Point(300.0, 0.3, 0.7)))", "M.VelocityModel().backward(s, o, DeltaPoint(Point(500.0, 0.6, 0.8), Point(300.0, 0.3, 0.9539392014169456)))",
        "M.ModelConvert(M.NoiseModel(), M.VelocityModel()).output_to(s, o, Point(500.0, 0.6, 0.8))", "M.ModelConvert(M.FlowModel(), M.DataModel()).output_from(s, o, Point(500.0, 0.6, 0.4))",
        "M.ModelConvert(M.VelocityModel(), M.VelocityModel()).output_to(s, o, Point(500.0, 0.6, 0.8))", "Point(613.0, 0.7391, 0.6733).add_noise(s, n)", "Point(613.0, 0.7391, 0.6733).remove_noise(s, n)",
        "Point(1000.0, 1.0, 0.0).remove_noise(s, n)",
    ]  # fmt: skip
    blob, meta = {}, []
    for dt in (torch.bfloat16, torch.float16):
        tag_dt = "bf16" if dt == torch.bfloat16 else "f16"
        g = torch.Generator().manual_seed(9000 + (dt == torch.float16))
        s_, o_, n_ = (torch.randn(2, 3, 8, 6, generator=g).to(dt) for _ in range(3))
        draws = [torch.randn(2, 3, 8, 6, generator=g).to(dt) for _ in range(40)]
        view = (lambda t: t.view(torch.int16).numpy().copy())
        blob[f"{tag_dt}/s"], blob[f"{tag_dt}/o"], blob[f"{tag_dt}/n"] = view(s_), view(o_), view(n_)
        blob[f"{tag_dt}/draws"] = np.stack([view(d) for d in draws])
        net = lambda xx, t, sg, al: xx * (0.3 - 0.1 * sg + 0.05 * al)  # noqa: E731
        for k, (stext, mtext, sched, steps_n) in enumerate(loops):
            pool = list(draws)
            res = eval(stext, names).sample_model(s_.clone(), net, eval(mtext, names), eval(sched, names), steps_n, rng=lambda *_: pool.pop(0))
            assert res.dtype == dt
            blob[f"{tag_dt}/loop{k}"] = view(res)
            meta.append({"key": f"{tag_dt}/loop{k}", "kind": "loop", "dtype": tag_dt, "sampler": stext, "model": mtext, "schedule": sched, "steps": steps_n, "draws_used": len(draws) - len(pool)})
        for k, text in enumerate(calls):
            res = eval(text, {**names, "s": s_, "o": o_, "n": n_})
            assert res.dtype == dt
            blob[f"{tag_dt}/call{k}"] = view(res)
            meta.append({"key": f"{tag_dt}/call{k}", "kind": "call", "dtype": tag_dt, "text": text})
    blob["meta"] = np.asarray(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "native_api16.npz"), **blob)


# ---------------------------------------------------------------------------------------------------
class _RecGen:
    "torch.Generator stand-in is impossible (C++ type); instead patch torch.randn/rand inside the noise module"


def noise() -> None:
    blob: dict = {}
    real_randn, real_rand = torch.randn, torch.rand

    def record(fn):
        normals, uniforms = [], []

        def randn(*a, **k):
            v = real_randn(*a, **k)
            normals.append(v.clone())
            return v

        def rand(*a, **k):
            v = real_rand(*a, **k)
            uniforms.append(v.item())
            return v

        RN.torch.randn, RN.torch.rand = randn, rand
        try:
            result = fn()
        finally:
            RN.torch.randn, RN.torch.rand = real_randn, real_rand
        return result, normals, uniforms

    def put(tag, result, normals, uniforms):
        blob[f"{tag}/out"] = result.numpy()
        blob[f"{tag}/uniforms"] = np.asarray(uniforms, dtype=np.float64)
        blob[f"{tag}/n_normals"] = np.asarray(len(normals))
        for i, n in enumerate(normals):
            blob[f"{tag}/normal{i}"] = n.numpy()

    g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731
    for unit in ((4, 16, 16), (4, 32, 24), (16, 16, 16)):
        u = "x".join(map(str, unit))
        gen = RN.Offset.from_inputs(unit, g(11))
        put(f"offset/{u}", *record(lambda: gen.generate(None)))
        gen = RN.Offset.from_inputs(unit, g(12), RN.OffsetProps(dims=(0, 2), strength=0.5))
        put(f"offset_d02/{u}", *record(lambda: gen.generate(None)))
        gen = RN.Pyramid.from_inputs(unit, g(13))
        put(f"pyramid/{u}", *record(lambda: gen.generate(None)))
        gen = RN.Pyramid.from_inputs(unit, g(14), RN.PyramidProps(strength=0.6, depth=1))
        put(f"pyramid_depth1/{u}", *record(lambda: gen.generate(None)))
        for j, st in enumerate((None, Step(0.0, 0.05), Step(0.45, 0.5), Step(0.95, 1.0))):
            gen = RN.Colored.from_inputs(unit, g(15 + j))
            put(f"colored{j}/{u}", *record(lambda: gen.generate(st)))
        gen = RN.Colored.from_inputs(unit, g(20), RN.ColoredProps(energy=2.5, color_start=1.5, color_end=-3, color_curve=0))
        put(f"colored_energy/{u}", *record(lambda: gen.generate(Step(0.3, 0.4))))
        blob[f"radial/{u}"] = RN.Colored._radial_freq_grid(torch.Size(unit), torch.device("cpu")).numpy()
    gen = RN.Pyramid.from_inputs((4, 128, 128), g(21))
    r, nn, uu = record(lambda: gen.generate(None))
    blob["pyramid_levels/4x128x128/uniforms"] = np.asarray(uu)
    blob["pyramid_levels/4x128x128/shapes"] = np.asarray([list(n.shape) for n in nn])
    blob["colored_exponents"] = np.asarray(
        [[a, b, _exp(Step(a, b))] for a, b in ((0, 0.05), (0.2, 0.25), (0.45, 0.5), (0.95, 1.0), (1.0, 0.9))], dtype=np.float64
    )
    np.savez_compressed(os.path.join(OUT, "noise.npz"), **blob)

    # noise_dims.npz: Pyramid over other `dims` subsets (noise.py:146-193 permutes the resized axes to the end, interpolates
    # slice by slice and permutes back; the level normals are drawn in the unit's own axis order)
    blob = {}
    for unit in ((4, 16, 24), (6, 10, 12)):
        u = "x".join(map(str, unit))
        for k, dims in enumerate(((0, 1), (0, 2), (0,), (1,), (-2,), (1, 2))):
            tag = "d" + "".join(str(d % len(unit)) for d in dims)
            gen = RN.Pyramid.from_inputs(unit, g(40 + k), RN.PyramidProps(dims=dims))
            try:
                put(f"pyramid_{tag}/{u}", *record(lambda: gen.generate(None)))
                blob[f"pyramid_{tag}/{u}/dims"] = np.asarray(dims)
            except Exception as exc:  # the reference itself rejects this choice: recorded, so the product may refuse it too
                blob[f"pyramid_{tag}/{u}/reference_error"] = np.asarray(type(exc).__name__)
    np.savez_compressed(os.path.join(OUT, "noise_dims.npz"), **blob)

    # colorize_nd.npz: Colored.colorize_noise (noise.py:337-403) on tensors with 4, 5 and 6 transform axes -- the whole tensor
    # is one sample there, so a batched video latent (B, C, T, H, W) has five
    blob = {}
    gw = torch.Generator().manual_seed(60)
    for shape, exponent, energy in (((3, 4, 5, 6, 8), 1.0, None), ((2, 3, 2, 5, 4, 6), -0.75, None), ((5, 1, 7, 3, 9, 10), 2.0, 1.7), ((6, 3, 10, 12), 0.5, None)):
        tag = "x".join(map(str, shape))
        white = torch.randn(shape, generator=gw)
        blob[f"{tag}/white"] = white.numpy()
        blob[f"{tag}/out"] = RN.Colored.colorize_noise(white.clone(), exponent, energy).numpy()
        blob[f"{tag}/args"] = np.asarray([exponent, float("nan") if energy is None else energy])
    np.savez_compressed(os.path.join(OUT, "colorize_nd.npz"), **blob)


# ---------------------------------------------------------------------------------------------------
def _norm(v):
    "JSON-able, class-identity-free rendering of config values (types by name, dataclasses by repr)"
    if isinstance(v, type):
        return f"<{v.__name__}>"
    if isinstance(v, dict):
        return {str(k): _norm(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_norm(x) for x in v]
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, torch.dtype):
        return str(v)
    return repr(v)


def wrapper_api() -> None:
    """The scheduler protocol around step(): set_timesteps in its four calling forms, timesteps / sigmas / init_noise_sigma / order /
    config, add_noise / scale_noise / scale_model_input / time_shift / set_begin_index on CPU tensors, the diffusers-config
    round trip (parse_diffusers_config, from_diffusers_config, as_diffusers_config) and the functional bridge."""
    out: dict = {"timesteps": {}, "scale": {}, "configs": {}, "functional": {}}
    mk = {
        "euler_scaled": lambda: RD.SkrampleWrapperScheduler(structured.Euler(), RS.Scaled()),
        "dpm2_karras": lambda: RD.SkrampleWrapperScheduler(structured.DPM(order=2, stochasticity=1), RS.Karras(RS.Scaled())),
        "unipc_flowshift": lambda: RD.SkrampleWrapperScheduler(structured.UniPC(order=3), RS.FlowShift(RS.Linear()), models.FlowModel()),
        "adams_zsnr_v": lambda: RD.SkrampleWrapperScheduler(structured.Adams(order=4), RS.ZSNR(), models.VelocityModel()),
        "euler_beta_flowshift": lambda: RD.SkrampleWrapperScheduler(structured.Euler(), RS.FlowShift(RS.Beta(RS.ZSNR()))),
        "euler_exp_static": lambda: RD.SkrampleWrapperScheduler(structured.Euler(), RS.Exponential(RS.Scaled()), allow_dynamic=False),
        "rku3_scaled": lambda: RD.RKUltraWrapperScheduler(RS.Scaled(), sampler_order=3),
        "dyn3_linear": lambda: RD.DynasauRKWrapperScheduler(RS.Linear(), sampler_order=3, model=models.FlowModel()),
    }
    forms = {
        "n7": dict(num_inference_steps=7),
        "n1": dict(num_inference_steps=1),
        "timesteps5": dict(timesteps=[900, 700, 500, 300, 100]),
        "sigmas4": dict(sigmas=[1.0, 0.7, 0.4, 0.1]),
        "n6_mu": dict(num_inference_steps=6, mu=0.8),
        "none": dict(),
    }
    g = torch.Generator().manual_seed(77)
    x = torch.randn([2, 3, 4, 4], generator=g, dtype=torch.float64)
    nz = torch.randn([2, 3, 4, 4], generator=g, dtype=torch.float64)
    out["x"], out["noise"] = x.flatten().tolist(), nz.flatten().tolist()
    for name, make in mk.items():
        for fname, kw in forms.items():
            w = make()
            w.set_timesteps(7)  # a previous run's state must not leak into the next call form
            rec: dict = {}
            try:
                w.set_timesteps(**kw)
                rec = {
                    "timesteps": w.timesteps.tolist(), "sigmas": w.sigmas.tolist(), "init_noise_sigma": float(w.init_noise_sigma),
                    "order": int(w.order), "schedule_np": w.schedule_np.tolist(), "config": _norm(dict(w.config)),
                    "schedule": repr(w.schedule),
                }  # fmt: skip
            except Exception as exc:
                rec = {"error": type(exc).__name__}
            out["timesteps"][f"{name}/{fname}"] = rec
        w = make()
        w.set_timesteps(6)
        ts = w.timesteps
        sc: dict = {"time_shift": [float(w.time_shift(0.7, 1.3, torch.tensor(t, dtype=torch.float64))) for t in (0.1, 0.5, 0.9)]}
        for k in (0, 2, len(ts) - 1):
            sc[f"scale_noise/{k}"] = w.scale_noise(x, ts[k], nz).flatten().tolist()
            sc[f"scale_model_input/{k}"] = w.scale_model_input(x, ts[k]).flatten().tolist()
            sc[f"scale_model_input_float/{k}"] = w.scale_model_input(x, float(ts[k])).flatten().tolist()
            sc[f"add_noise/{k}"] = w.add_noise(x, nz, ts[k : k + 2]).flatten().tolist()
        sc["add_noise/empty"] = w.add_noise(x, nz, ts[:0]).flatten().tolist()
        try:
            w.set_begin_index(2 * w.order)
            sc["begin_index/config"] = _norm(dict(w.config)).get("begin_index")
            sc["begin_index/add_noise"] = w.add_noise(x, nz, ts[2 * w.order : 2 * w.order + 1]).flatten().tolist()
        except Exception as exc:
            sc["begin_index/error"] = type(exc).__name__
        out["scale"][name] = sc

    # diffusers config round trip
    base_cfgs = []
    for cls_name in RD.DIFFUSERS_CLASS_MAP:
        base_cfgs.append({"_class_name": cls_name})
    variants = [
        {},
        {"prediction_type": "v_prediction", "beta_schedule": "scaled_linear", "use_karras_sigmas": True, "solver_order": 3},
        {"prediction_type": "sample", "beta_schedule": "linear", "use_exponential_sigmas": True, "algorithm_type": "sde-dpmsolver++", "num_train_timesteps": 500},
        {"prediction_type": "epsilon", "use_beta_sigmas": True, "rescale_betas_zero_snr": True, "timestep_spacing": "trailing"},
        {"prediction_type": "flow", "shift": 3.0, "use_dynamic_shifting": False},
        {"flow_shift": 2.0, "use_flow_sigmas": True, "solver_order": 2, "algorithm_type": "dpmsolver++"},
        {"shift": 1.5, "use_dynamic_shifting": True, "base_shift": 0.5, "max_shift": 1.15, "use_karras_sigmas": True},
    ]
    for bc in base_cfgs:
        for vi, var in enumerate(variants):
            cfg = {**bc, **var}
            key = f"{bc['_class_name']}/{vi}"
            rec = {"config": cfg}
            try:
                parsed = RD.parse_diffusers_config(cfg)
                rec["parsed"] = _norm(dataclasses.asdict(parsed) if False else {f.name: getattr(parsed, f.name) for f in dataclasses.fields(parsed)})
                w = RD.SkrampleWrapperScheduler.from_diffusers_config(cfg)
                rec["wrapper"] = {"sampler": repr(w.sampler), "schedule": repr(w.schedule), "model": repr(w.model), "invert": bool(w.invert_prediction)}
                rec["as_config"] = _norm(RD.as_diffusers_config(w.sampler, w.schedule, w.model))
                w.set_timesteps(5)
                rec["timesteps"] = w.timesteps.tolist()
                rec["sigmas"] = w.sigmas.tolist()
            except Exception as exc:
                rec["error"] = type(exc).__name__
            out["configs"][key] = rec
    # explicit sampler / schedule arguments override what the config names
    for key, kw in {
        "override_sampler": dict(sampler=structured.Adams),
        "override_schedule": dict(schedule=RS.Linear),
    }.items():
        cfg = {"_class_name": "DPMSolverMultistepScheduler", "solver_order": 3, "prediction_type": "epsilon"}
        parsed = RD.parse_diffusers_config(cfg, **kw)
        out["configs"][key] = {"config": cfg, "parsed": _norm({f.name: getattr(parsed, f.name) for f in dataclasses.fields(parsed)})}

    # functional bridge on CPU float64 tensors with a linear toy model
    def toy(xx, t, s, a):
        return xx * 0.3 - 0.1 * s + 0.05 * a

    for name in ("euler_scaled", "dpm2_karras", "adams_zsnr_v", "rku3_scaled"):
        w = mk[name]()
        draws = [torch.randn([2, 3, 4, 4], generator=g, dtype=torch.float64) for _ in range(24)]
        pool = list(draws)
        res = w.functional_sample_model(x.clone(), toy, 5, rng=lambda *_: pool.pop(0))
        used = len(draws) - len(pool)
        pool2 = list(draws)
        gen = w.functional_generate_model(toy, lambda *_: pool2.pop(0), 5)
        out["functional"][name] = {
            "draws": [d.flatten().tolist() for d in draws[: max(used, len(draws) - len(pool2))]],
            "sample_model": res.flatten().tolist(), "used": used,
            "generate_model": gen.flatten().tolist(), "used_generate": len(draws) - len(pool2),
        }  # fmt: skip
    json.dump(out, open(os.path.join(OUT, "wrapper_api.json"), "w"))


# ---------------------------------------------------------------------------------------------------
FUNCTIONAL_CASES = {
    # name: (sampler factory as text evaluated in the test too, model name, schedule name, steps, include)
    "rku1_eps": ("F.RKUltra(order=1)", "eps", "scaled", 6, (None, None)),
    "rku2_flow": ("F.RKUltra(order=2)", "flow", "linear", 6, (None, None)),
    "rku3_sde_eps": ("F.RKUltra(order=3, stochasticity=0.5)", "eps", "karras_scaled", 6, (None, None)),
    "rku4_v_slice": ("F.RKUltra(order=4)", "v", "zsnr", 7, (2, 5)),
    "rku6_sde_flow": ("F.RKUltra(order=6, stochasticity=1)", "flow", "linear", 4, (None, None)),
    "rku5_noderiv": ("F.RKUltra(order=5, derivative_transform=None)", "eps", "scaled", 4, (None, None)),
    "dyn2_eps": ("F.DynasauRK(order=2)", "eps", "scaled", 6, (None, None)),
    "dyn3_flow_invert": ("F.DynasauRK(order=3, invert=True)", "flow", "linear", 5, (None, None)),
    "dyn4_decay": ("F.DynasauRK(order=4, per_step_decay=0.1, total_step_decay=-0.02, stochasticity=0.3)", "eps", "scaled", 5, (1, None)),
    "moire2": ("F.RKMoire(order=2)", "eps", "scaled", 6, (None, None)),
    "moire3_flow": ("F.RKMoire(order=3, threshold=1e-3, initial=1 / 20)", "flow", "linear", 6, (None, None)),
    "moire5_mae": ("F.RKMoire(order=5, evaluator=F.FunctionalAdaptive.mae, adaption=0.5, rescale_max=True)", "eps", "scaled", 8, (None, None)),
    "moire2_slice_discard": ("F.RKMoire(order=2, discard=2.0, maximum=1 / 3)", "v", "zsnr", 8, (1, 6)),
    "adapter_dpm2_sde": ("I.StructuredFunctionalAdapter(S.DPM(order=2, stochasticity=1))", "eps", "karras_scaled", 7, (None, None)),
    "adapter_unipc3_slice": ("I.StructuredFunctionalAdapter(S.UniPC(order=3))", "flow", "linear", 8, (2, 7)),
    "adapter_spc": ("I.StructuredFunctionalAdapter(S.SPC())", "eps", "scaled", 6, (None, None)),
}


def functional_api() -> None:
    """The functional samplers (RKUltra, DynasauRK, adaptive RKMoire, the structured adapter) driving a toy model on CPU float64
    tensors: the result, every (t, sigma, alpha) the model was called at, the callback trace and the draws consumed."""
    from skrample.sampling import interface

    env = {"F": functional, "S": structured, "I": interface}
    g = torch.Generator().manual_seed(4242)
    x = torch.randn([2, 3, 4], generator=g, dtype=torch.float64)
    out: dict = {"x": x.flatten().tolist(), "cases": {}}
    for name, (expr, mname, sname, steps, (lo, hi)) in FUNCTIONAL_CASES.items():
        sampler = eval(expr, env)
        sched, model_t = CFG_SCHEDULES[sname](), MODELS[mname]
        draws = [torch.randn([2, 3, 4], generator=g, dtype=torch.float64) for _ in range(40)]
        pool = list(draws)
        seen, trace = [], []

        def toy(xx, t, s, a):
            seen.append([float(t), float(s), float(a)])
            return xx * 0.3 - 0.1 * s + 0.05 * a + 0.01 * torch.sin(xx * 3.0)

        def cb(sample, n, dp):
            trace.append([int(n), *[float(v) for v in dp.point_from], *[float(v) for v in dp.point_to], float(sample.sum())])

        rec: dict = {"expr": expr, "model": mname, "schedule": sname, "steps": steps, "include": [lo, hi], "adjust_steps": sampler.adjust_steps(steps) if hasattr(sampler, "adjust_steps") else None}
        try:
            res = sampler.sample_model(x.clone(), toy, model_t, sched, steps, slice(lo, hi), lambda *_: pool.pop(0), cb)
            rec.update(result=res.flatten().tolist(), seen=list(seen), trace=list(trace), used=len(draws) - len(pool))
            rec["draws"] = [d.flatten().tolist() for d in draws[: rec["used"]]]
            seen.clear(); trace.clear()
            pool2 = list(draws)
            gen = sampler.generate_model(toy, model_t, sched, lambda *_: pool2.pop(0), steps, slice(lo, hi), None if lo is None else x.clone())
            rec.update(generate=gen.flatten().tolist(), generate_used=len(draws) - len(pool2), generate_nfe=len(seen))
            rec["generate_draws"] = [d.flatten().tolist() for d in draws[: rec["generate_used"]]]
        except Exception as exc:
            rec["error"] = type(exc).__name__
        out["cases"][name] = rec
    json.dump(out, open(os.path.join(OUT, "functional_api.json"), "w"))


# ---------------------------------------------------------------------------------------------------
NATIVE16_CASES = {
    # name: (sampler text, model, schedule)
    "euler": ("S.Euler()", "eps", "scaled"),
    "euler_sde_v": ("S.Euler(stochasticity=1)", "v", "zsnr"),
    "dpm2_sde": ("S.DPM(order=2, stochasticity=1)", "eps", "karras_scaled"),
    "dpm3_flow": ("S.DPM(order=3)", "flow", "linear"),
    "adams3": ("S.Adams(order=3)", "eps", "scaled"),
    "unipc3_flow": ("S.UniPC(order=3)", "flow", "linear"),
}


def native16() -> None:
    """The sampler-level API called directly on 16-bit tensors (no wrapper): the reference then computes in the TENSOR dtype,
    rounding after every torch op (structured.py:209-283).  Six samplers x {bf16, fp16}, 7 teacher-forced steps each."""
    env = {"S": structured}
    blob = {}
    for ci, (name, (expr, mname, sname)) in enumerate(NATIVE16_CASES.items()):
        for dt in (torch.bfloat16, torch.float16):
            sampler = eval(expr, env)
            sched, model_t = CFG_SCHEDULES[sname](), MODELS[mname]
            g = torch.Generator().manual_seed(5000 + ci)
            steps, shape = 7, (2, 4, 8, 8)
            x = torch.randn(shape, generator=g).to(dt)
            previous: list = []
            tag = f"{name}/{'bf16' if dt == torch.bfloat16 else 'f16'}"
            xs, outs, nzs, finals, preds = [], [], [], [], []
            for i in range(steps):
                out = torch.randn(shape, generator=g).to(dt)
                nz = torch.randn(shape, generator=g).to(dt)
                rec = sampler.sample(x, out, Step.from_int(i, steps), model_t, sched, nz if sampler.require_noise else None, tuple(previous))
                assert rec.final.dtype == dt
                xs.append(x); outs.append(out); nzs.append(nz); finals.append(rec.final); preds.append(rec.prediction)
                previous.append(rec)
                previous = previous[max(len(previous) - sampler.require_previous, 0) :] if sampler.require_previous else []
                x = rec.final
            as16 = lambda ts: np.stack([t.contiguous().view(torch.int16).numpy().copy() for t in ts])  # noqa: E731
            blob[f"{tag}/x"], blob[f"{tag}/out"], blob[f"{tag}/noise"] = as16(xs), as16(outs), as16(nzs)
            blob[f"{tag}/final"], blob[f"{tag}/prediction"] = as16(finals), as16(preds)
            blob[f"{tag}/meta"] = np.asarray([expr, mname, sname, str(steps)])
    np.savez_compressed(os.path.join(OUT, "native16.npz"), **blob)


# ---------------------------------------------------------------------------------------------------
def common_api() -> None:
    "skrample/common.py's helpers on floats, numpy arrays and tensors, edge values included (common.py:24-213)"
    import skrample.common as C

    def num(v):
        if isinstance(v, (tuple, list)):
            return [num(x) for x in v]
        if isinstance(v, torch.Tensor):
            return num(v.tolist())
        if isinstance(v, np.ndarray):
            return num(v.tolist())
        v = float(v)
        return v if math.isfinite(v) else repr(v)

    def attempt(fn):
        try:
            return num(fn())
        except Exception as exc:
            return {"error": type(exc).__name__}

    xs = [-3.5, -1.0, -0.25, -0.0, 0.0, 1e-12, 0.5, 1.0, 2.0, 40.0, 1e30, float("inf")]
    arr = np.asarray([-2.0, -0.5, 0.0, 0.3, 1.7])
    ten = torch.tensor(arr)
    out: dict = {"xs": num(xs), "arr": arr.tolist()}
    out["divf"] = [[attempt(lambda a=a, b=b: C.divf(a, b)) for b in (-2.0, -0.0, 0.0, 3.0)] for a in (-1.5, 0.0, 2.0)]
    out["ln"] = [attempt(lambda x=x: C.ln(x)) for x in xs]
    out["rescale_positive"] = [attempt(lambda x=x: C.rescale_positive(x)) for x in xs[:-1]]
    out["rescale_subnormal"] = [attempt(lambda x=x: C.rescale_subnormal(x)) for x in xs]
    out["exp"] = [attempt(lambda x=x: C.exp(x)) for x in xs[:10]] + [attempt(lambda: C.exp(arr)), attempt(lambda: C.exp(ten))]
    out["sigmoid"] = [attempt(lambda x=x: C.sigmoid(x)) for x in xs[:10]] + [attempt(lambda: C.sigmoid(arr)), attempt(lambda: C.sigmoid(ten))]
    out["softmax"] = [attempt(lambda: C.softmax((0.1, -2.0, 3.0))), attempt(lambda: C.softmax((arr, arr * 0.5, arr - 1))), attempt(lambda: C.softmax((ten, ten * 2)))]
    out["spowf"] = [[attempt(lambda x=x, f=f: C.spowf(x, f)) for f in (0.5, 1.0, 2.0, -1.0)] for x in xs[:10]] + [[attempt(lambda f=f: C.spowf(arr, f)), attempt(lambda f=f: C.spowf(ten, f))] for f in (0.5, 2.0)]
    out["mean"] = [attempt(lambda: C.mean(0.75)), attempt(lambda: C.mean(arr)), attempt(lambda: C.mean(ten))]
    out["clamp"] = [attempt(lambda x=x: C.clamp(x)) for x in xs] + [attempt(lambda: C.clamp(5.0, 2, 3)), attempt(lambda: C.clamp(-5.0, -1, 3))]
    out["normalize"] = [attempt(lambda: C.normalize(0.3, 2.0)), attempt(lambda: C.normalize(arr, 4.0, 1.0)), attempt(lambda: C.normalize(ten, 0.5, -0.5))]
    out["regularize"] = [attempt(lambda: C.regularize(0.3, 2.0)), attempt(lambda: C.regularize(arr, 4.0, 1.0)), attempt(lambda: C.regularize(ten, 0.5, -0.5))]
    out["bashforth"] = [attempt(lambda n=n: C.bashforth(n)) for n in range(1, 10)]
    p = C.Point(700.0, 1.3, 0.4)
    p0 = C.Point(0.0, 0.0, 1.0)
    pz = C.Point(999.0, 1.0, 0.0)
    out["point"] = {
        "add_noise": [attempt(lambda: p.add_noise(0.5, -2.0)), attempt(lambda: p.add_noise(arr, arr[::-1].copy())), attempt(lambda: p.add_noise(ten, ten * 3))],
        "remove_noise": [attempt(lambda: p.remove_noise(0.5, -2.0)), attempt(lambda: p.remove_noise(arr, arr[::-1].copy())), attempt(lambda: pz.remove_noise(0.5, -2.0)), attempt(lambda: p0.remove_noise(ten, ten * 3))],
        "difference": attempt(lambda: C.DeltaPoint(p, p0).difference()),
    }
    steps = []
    for n, amount in ((0, 1), (0, 7), (3, 7), (6, 7), (19, 20)):
        st = C.Step.from_int(n, amount)
        steps.append({
            "from_int": [n, amount], "step": num(st), "distance": attempt(st.distance), "position": attempt(st.position), "amount": attempt(st.amount),
            "offset": [attempt(lambda k=k: st.offset(k)) for k in (-2, 0.5, 3)], "clamp": [attempt(lambda k=k: st.offset(k).clamp()) for k in (-9, 0, 9.5)],
            "normal": attempt(lambda: C.Step(st.time_to, st.time_from).normal()),
        })  # fmt: skip
    out["step"] = steps
    a, b = list(range(0, 11)), list(range(0, 15, 2))
    out["merge"] = {m.name: {"value": str(m.value), "ab": m.merge(a, b), "ba": m.merge(b, a), "cmp": m.merge(a, b, lambda u, v: u // 2 == v // 2)} for m in C.MergeStrategy}
    json.dump(out, open(os.path.join(OUT, "common_api.json"), "w"))


# ---------------------------------------------------------------------------------------------------
SCHEDULE_EXPRS = [
    "R.Scaled()", "R.Scaled(base_timesteps=500, beta_start=0.0001, beta_end=0.02, beta_scale=1)", "R.Scaled(beta_scale=3)", "R.ZSNR()",
    "R.ZSNR(beta_scale=1, base_timesteps=200)", "R.Linear()", "R.Linear(sigma_start=14.6)", "R.Linear(base_timesteps=250, sigma_start=0.8)",
    "R.Karras(R.Scaled())", "R.Karras(R.Scaled(), rho=3.0, steps=9)", "R.Karras(R.Linear())", "R.Karras(R.ZSNR(), steps=31)",
    "R.Exponential(R.Scaled())", "R.Exponential(R.Scaled(), rho=2.5, steps=7)", "R.Exponential(R.Linear())",
    "R.Beta(R.Scaled())", "R.Beta(R.ZSNR(), alpha=0.4, beta=1.3)", "R.Beta(R.Linear(), alpha=2.0, beta=0.5)",
    "R.Probit(R.Linear())", "R.Probit(R.Scaled(), scale=1.5)", "R.NoSub(R.Scaled())",
    "R.FlowShift(R.Linear())", "R.FlowShift(R.Linear(), shift=0.5)", "R.FlowShift(R.Scaled(), shift=7.0)", "R.FlowShift(R.Beta(R.ZSNR()))",
    "R.Hyper(R.Scaled())", "R.Hyper(R.Linear(), scale=-1.5, tail=False)", "R.Hyper(R.Karras(R.Scaled()), scale=0.5)",
    "R.Sinner(R.Linear())", "R.Sinner(R.Scaled(), count=3, scale=-0.7)", "R.NoMod(R.Linear())",
    "R.FlowShift(R.Hyper(R.Sinner(R.Karras(R.Scaled()))), shift=2.0)", "R.Hyper(R.FlowShift(R.Probit(R.Linear()), shift=1.7), scale=3)",
]


def scheduling_api() -> None:
    """Every schedule class, sub-schedule and modifier (defaults and non-default parameters, nested stacks): schedule_np for four
    run lengths, points / ipoints at fixed abscissae, step / istep, the end points, the sigma space, the modifier-stack
    introspection (all_split, lowest, find, find_split, stack round trip)."""
    env = {"R": RS}
    ts = [0.0, 1e-9, 0.013, 0.25, 0.5, 0.77, 0.999, 1.0]
    out: dict = {"t": ts, "cases": {}}
    for expr in SCHEDULE_EXPRS:
        sch = eval(expr, env)
        rec: dict = {"repr": repr(sch), "space": type(sch.space).__name__}
        for n in (1, 2, 9, 30):
            rec[f"schedule_np/{n}"] = sch.schedule_np(n).tolist()
        rec["points"] = [list(map(float, p)) for p in sch.points(ts)]
        rec["ipoints"] = [list(map(float, p)) for p in sch.ipoints(ts)]
        rec["point_0"], rec["point_1"] = list(map(float, sch.point_0)), list(map(float, sch.point_1))
        st = Step.from_int(2, 9)
        rec["step"] = [list(map(float, p)) for p in sch.step(st)]
        rec["istep"] = [list(map(float, p)) for p in sch.istep(st)]
        if isinstance(sch, RS.ScheduleModifier):
            mods, sub, base = sch.all_split
            rec["all_split"] = [[repr(m) for m in mods], repr(sub), repr(base)]
            rec["lowest"] = repr(sch.lowest)
            rec["all"] = [repr(v) for v in sch.all]
            rec["find_flowshift"] = repr(sch.find(RS.FlowShift))
            rec["find_hyper_exact"] = repr(sch.find(RS.Hyper, exact=True))
            found = sch.find_split(RS.FlowShift)
            rec["find_split_flowshift"] = None if found is None else [[repr(m) for m in found[0]], repr(found[1]), [repr(m) for m in found[2]], repr(found[3]), repr(found[4])]
            rec["restacked"] = repr(sch.stack(mods, sub, base))
        out["cases"][expr] = rec
    # a fixed table of (timestep, sigma) pairs behaves as a schedule too
    fixed = RS.FixedSchedule.from_regular(np.asarray([900.0, 600.0, 300.0, 50.0]), np.asarray([10.0, 3.0, 0.8, 0.05]), RS.VariancePreserving())
    out["fixed"] = {"repr": repr(fixed), "schedule_np/4": fixed.schedule_np(4).tolist(), "points": [list(map(float, p)) for p in fixed.points([0.0, 0.3, 1.0])]}
    for name, space in (("vp", RS.VariancePreserving()), ("flow", RS.FlowMatching())):
        sig = np.asarray([0.0, 0.05, 0.7, 1.0, 14.6])
        try:
            out[f"space/{name}"] = {"normalize": [np.asarray(a).tolist() for a in space.normalize(sig)], "regularize": np.asarray(space.regularize(np.asarray([0.0, 0.05, 0.5, 0.9]))).tolist()}
        except Exception as exc:
            out[f"space/{name}"] = {"error": type(exc).__name__}
    sch = RS.Karras(RS.Scaled())
    out["lru"] = {"np": RS.np_schedule_lru(sch, 6).tolist(), "points": [list(map(float, p)) for p in RS.schedule_lru(sch, 6)]}
    json.dump(out, open(os.path.join(OUT, "scheduling_api.json"), "w"))


# ---------------------------------------------------------------------------------------------------
MODEL_EXPRS = ["M.DataModel()", "M.NoiseModel()", "M.FlowModel()", "M.VelocityModel()", "M.ScaleX()", "M.ScaleX(bias=-1.5)"]


def models_api() -> None:
    """models.py on CPU float64 tensors and floats: to_x / from_x, gamma / delta / zeta / zeta_ts / eta_transform, forward /
    backward with and without noise, ModelConvert between every pair (output_to, output_from, wrap_model_call)."""
    env = {"M": models}
    g = torch.Generator().manual_seed(99)
    x, o, nz = (torch.randn([2, 3], generator=g, dtype=torch.float64) for _ in range(3))
    out: dict = {"x": x.flatten().tolist(), "o": o.flatten().tolist(), "noise": nz.flatten().tolist(), "cases": {}, "convert": {}}
    scheds = {"scaled": RS.Scaled(), "linear": RS.Linear(), "zsnr": RS.ZSNR()}
    spans = [(0.0, 0.1), (0.45, 0.5), (0.3, 0.8), (0.9, 1.0)]

    def num(v):
        v = float(v)
        return v if math.isfinite(v) else repr(v)

    def attempt(fn):
        try:
            r = fn()
            if isinstance(r, torch.Tensor):
                return [num(v) for v in r.flatten().tolist()]
            if isinstance(r, tuple):
                return [[num(q) for q in p] for p in r]
            return num(r)
        except Exception as exc:
            return {"error": type(exc).__name__}

    for expr in MODEL_EXPRS:
        m = eval(expr, env)
        for sname, sch in scheds.items():
            for a, b in spans:
                dp = DeltaPoint(*sch.ipoints([a, b]))
                key = f"{expr}|{sname}|{a}|{b}"
                rec = {"repr": repr(m)}
                rec["to_x"] = attempt(lambda: m.to_x(x, o, dp.point_from))
                rec["from_x"] = attempt(lambda: m.from_x(x, o, dp.point_from))
                rec["to_x_float"] = attempt(lambda: m.to_x(0.7, -0.4, dp.point_from))
                for eta in (0.0, 0.5, 1.0, -1.5):
                    rec[f"gdz/{eta}"] = [attempt(lambda: m.gamma(dp, eta)), attempt(lambda: m.delta(dp, eta)), attempt(lambda: m.zeta(dp, eta)), attempt(lambda: m.zeta_ts(dp, eta))]
                    rec[f"eta_transform/{eta}"] = attempt(lambda: m.eta_transform(dp, eta))
                    rec[f"forward/{eta}"] = attempt(lambda: m.forward(x, o, dp, nz, eta))
                    rec[f"backward/{eta}"] = attempt(lambda: m.backward(x, o, dp, nz, eta))
                rec["forward/plain"] = attempt(lambda: m.forward(x, o, dp))
                rec["backward/plain"] = attempt(lambda: m.backward(x, o, dp))
                rec["forward/float"] = attempt(lambda: m.forward(0.7, -0.4, dp, 0.2, 1.0))
                out["cases"][key] = rec
    pt = RS.Scaled().ipoint(0.4)
    for ea in MODEL_EXPRS:
        for eb in MODEL_EXPRS:
            cv = models.ModelConvert(eval(ea, env), eval(eb, env))
            wrapped = cv.wrap_model_call(lambda xx, t, s, a: xx * 0.3 - 0.1 * s + 0.05 * a)
            out["convert"][f"{ea}->{eb}"] = {
                "output_to": attempt(lambda: cv.output_to(x, o, pt)), "output_from": attempt(lambda: cv.output_from(x, o, pt)),
                "wrapped": attempt(lambda: wrapped(x, *pt)),
            }  # fmt: skip
    json.dump(out, open(os.path.join(OUT, "models_api.json"), "w"))


def _exp(step) -> float:
    seen = {}
    orig = RN.Colored.colorize_noise
    RN.Colored.colorize_noise = staticmethod(lambda white, exponent=0.0, energy=None: seen.setdefault("e", exponent) and white)
    try:
        RN.Colored.from_inputs((2, 2), torch.Generator().manual_seed(0)).generate(step)
    finally:
        RN.Colored.colorize_noise = orig
    return seen["e"]


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    kats()
    tables()
    steps()
    sweep()
    sweep_native()
    native_api16()
    noise()
    wrapper_api()
    functional_api()
    native16()
    common_api()
    scheduling_api()
    models_api()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
