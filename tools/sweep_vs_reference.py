"""Build-container check (needs /root/reference): draw N random wrapper configurations from tools/make_golden.py's sweep grammar,
step each through the imported reference and through skrample_amd on host tensors with the same inputs (teacher-forced), and report
every difference -- results beyond the parity bar, different timesteps, or one side raising where the other does not.
tests/golden/steps_sweep.npz holds the first 64 accepted cases of the same generator; this runs as many as asked.

    python tools/sweep_vs_reference.py [count] [first_seed] [wrapper|functional|noise|config|sampler|schedule|generator|native|array|model]
"""

import math
import os
import random
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
warnings.filterwarnings("ignore")

import make_golden as MG  # noqa: E402  (installs the reference loader)

import skrample_amd.diffusers as PD  # noqa: E402
import skrample_amd.scheduling as PS  # noqa: E402
from skrample_amd.sampling import lazy  # noqa: E402
from skrample_amd.sampling import models as PM  # noqa: E402
from skrample_amd.sampling import structured as PT  # noqa: E402

def bf16_ulp(ref: torch.Tensor) -> torch.Tensor:
    mag = ref.float().abs().clamp_min(2.0**-126)
    return torch.exp2(torch.floor(torch.log2(mag)) - 7)


NOISE_LIKE_SAMPLE = False  # the "native" mode of __main__ sets it: compute_scale=None wrappers on 16-bit tensors

REF = {"W": MG.RD, "T": MG.structured, "S": MG.RS, "M": MG.models, "torch": torch}
OWN = {"W": PD, "T": PT, "S": PS, "M": PM, "torch": torch}


class _Replay:
    def __init__(self, draws):
        self.draws = list(draws)

    def generate(self, step):
        return self.draws.pop(0)

    generate_lazy = generate


def one(seed: int) -> str | None:
    text, dtype, shape, steps_n = MG._sweep_spec(random.Random(seed))
    dt = getattr(torch, dtype)
    g = torch.Generator().manual_seed(seed)
    try:
        r = eval(text, REF)
        r.set_timesteps(steps_n)
        r_times = r.timesteps.clone()
        r_err = None
    except Exception as err:  # noqa: BLE001
        r_err = err
    try:
        p = eval(text, OWN)
        p.set_timesteps(steps_n)
        p_times = p.timesteps.clone()
        p_err = None
    except Exception as err:  # noqa: BLE001
        p_err = err
    if r_err or p_err:
        if type(r_err) is not type(p_err):
            return f"set_timesteps: reference {r_err!r}, here {p_err!r}"
        return None
    if r_times.shape != p_times.shape or (r_times - p_times).abs().max() > 1e-9:
        return f"timesteps differ: {r_times.tolist()} vs {p_times.tolist()}"
    x = torch.randn(shape, generator=g).to(dt)
    n = len(r_times)
    outs = [torch.randn(shape, generator=g).to(dt) for _ in range(n)]
    noises = [torch.randn(shape, generator=g).to(dt if NOISE_LIKE_SAMPLE else torch.float32) for _ in range(n)]  # (what get_step_noise hands over is cast to compute_scale, or to the sample's dtype when that is None)
    # the helper entry points a diffusers pipeline calls around the loop (img2img start: add_noise / scale_noise at a later timestep,
    # set_begin_index, scale_model_input, init_noise_sigma), then the steps from that start with an empty history
    extras = random.Random(seed ^ 0x5EED)
    stride = max(int(getattr(r, "order", 1)), 1)
    start = extras.randrange(0, n, stride) if extras.random() < 0.4 else 0
    probes = []
    for w, times in ((r, r_times), (p, p_times)):
        try:
            if start:
                w.set_begin_index(start)
            probe = [float(w.init_noise_sigma)]
            probe.append(torch.as_tensor(w.add_noise(x, noises[0].to(dt), times[start : start + 1])).double())
            probe.append(torch.as_tensor(w.scale_noise(x, times[start], noises[0].to(dt))).double())
            probe.append(torch.as_tensor(w.scale_model_input(x, times[start])).double())
            probes.append(probe)
        except Exception as err:  # noqa: BLE001
            probes.append(err)
    if isinstance(probes[0], Exception) or isinstance(probes[1], Exception):
        return None if type(probes[0]) is type(probes[1]) else f"helper entry points: reference {probes[0]!r}, here {probes[1]!r}"
    if abs(probes[0][0] - probes[1][0]) > 1e-12 * max(abs(probes[0][0]), 1.0):
        return f"init_noise_sigma {probes[0][0]} vs {probes[1][0]}"
    for name, a, b in zip(("add_noise", "scale_noise", "scale_model_input"), probes[1][1:], probes[0][1:]):
        exact = dt in (torch.bfloat16, torch.float16)  # (16-bit: the reference's own rounded ops, bit for bit)
        if a.shape != b.shape or not (torch.equal(a, b) if exact else torch.allclose(a, b, rtol=1e-5, atol=1e-6, equal_nan=True)):
            return f"{name} at timestep index {start} differs: max diff {(a - b).abs().max().item():.3g}"
    for lap in range(LAPS):
        if lap:  # the same two objects again (SWEEP_LAPS=2): a full run on fresh inputs behind the first one, whatever state that left
            r.set_timesteps(steps_n)
            p.set_timesteps(steps_n)
            start = 0
            x = torch.randn(shape, generator=g).to(dt)
            outs = [torch.randn(shape, generator=g).to(dt) for _ in range(n)]
            noises = [torch.randn(shape, generator=g).to(dt if NOISE_LIKE_SAMPLE else torch.float32) for _ in range(n)]
        found = _steps(r, p, r_times, p_times, start, x, outs, noises, dt)
        if found:
            return found if not lap else f"second run: {found}"
    return None


LAPS = int(os.environ.get("SWEEP_LAPS", "1"))


def _steps(r, p, r_times, p_times, start, x, outs, noises, dt) -> str | None:
    r._noise_generator, p._noise_generator = MG._Injected(noises[start:]), _Replay(noises[start:])
    for i, (tr, tp) in enumerate(zip(r_times, p_times)):
        if i < start:
            continue
        try:
            ref = r.step(outs[i], tr, x, return_dict=False)
            r_err = None
        except Exception as err:  # noqa: BLE001
            r_err = err
        try:
            got = p.step(outs[i], tp, x, return_dict=False)
            got = [torch.as_tensor(v.materialize() if isinstance(v, lazy.LazyTensor) else v) for v in got]
            p_err = None
        except Exception as err:  # noqa: BLE001
            p_err = err
        if r_err or p_err:
            if r_err and p_err:
                return None
            finite = r_err is None and all(torch.isfinite(v.float()).all() for v in ref)
            if p_err and not finite:
                return None  # (singular point: the reference hands back inf / nan, the engine raises -- INTEGRATION.md)
            return f"step {i}: reference {r_err!r}, here {p_err!r}"
        if not all(torch.isfinite(v.float()).all() for v in ref):
            return None
        for name, a, b in zip(("prev", "pred"), got, ref):
            if a.dtype != b.dtype or a.shape != b.shape:
                return f"step {i} {name}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}"
            a64, b64 = a.double(), b.double()
            scale = b64.abs().max().clamp_min(1e-30)
            if dt in (torch.float32, torch.float64):
                allowed = 1e-5 * scale  # north_star's bar: relative inf-norm error
            else:  # one unit in the last place of the reference's 16-bit result (+ 1e-6 max|ref| for cancellation residues), as tests/test_step_gpu.py
                allowed = bf16_ulp(b).double() * (1 if dt == torch.bfloat16 else 2.0**-3) + 1e-6 * scale
            bad = ((a64 - b64).abs() > allowed).sum().item()
            if bad:
                return f"step {i} {name}: {bad} elements beyond the bar (max diff {(a64 - b64).abs().max().item():.3g}, scale {scale.item():.3g})"
        x = ref[0]
    return None


# ---- the functional samplers (RKUltra, DynasauRK, adaptive RKMoire, the structured adapter) on float64 host tensors ----------
from sweep_grammar import functional_spec as _functional_spec  # noqa: E402  (tests/sweep_grammar.py: shared with tests/soak_functional.py on the GPU box)


def one_functional(seed: int) -> str | None:
    import skrample.sampling.functional as RF
    import skrample.sampling.interface as RI

    import skrample_amd.sampling.functional as OF
    import skrample_amd.sampling.interface as OI

    text, schedule, model, steps_n, (lo, hi) = _functional_spec(random.Random(seed))
    g = torch.Generator().manual_seed(seed)
    kind = random.Random(seed ^ 0xA11).choice(("tensor", "tensor", "f64", "f32", "float", "bf16", "f16"))  # what the reference's generic `T` covers
    make = {"tensor": lambda: torch.randn([2, 3, 4], generator=g, dtype=torch.float64), "f64": lambda: torch.randn([2, 3, 4], generator=g, dtype=torch.float64).numpy(),
            "bf16": lambda: torch.randn([2, 3, 4], generator=g).bfloat16(), "f16": lambda: torch.randn([2, 3, 4], generator=g).half(),
            "f32": lambda: torch.randn([2, 3, 4], generator=g).numpy(), "float": lambda: float(torch.randn([], generator=g, dtype=torch.float64))}[kind]  # fmt: skip
    x = make()
    draws = [make() for _ in range(400)]
    sides = []
    for names, F, I in ((REF, RF, RI), (OWN, OF, OI)):
        env = {**names, "F": F, "I": I}
        seen, trace, pool = [], [], list(draws)

        def toy(xx, t, s, a):
            seen.append((float(t), float(s), float(a)))
            return xx * 0.3 - 0.1 * s + 0.05 * a + (0.01 * torch.sin(xx * 3.0) if isinstance(xx, torch.Tensor) else 0.01 * np.sin(xx * 3.0))

        def cb(sample, n, dp):
            trace.append((int(n), *[float(v) for v in dp.point_from], *[float(v) for v in dp.point_to], float((sample.double() if isinstance(sample, torch.Tensor) else np.asarray(sample, dtype=np.float64)).sum())))

        try:
            sampler = eval(text, env)
            res = sampler.sample_model(x.clone() if isinstance(x, torch.Tensor) else (x.copy() if isinstance(x, np.ndarray) else x), toy, eval(model, env), eval(schedule, env), steps_n, slice(lo, hi), lambda *_: pool.pop(0), cb)
            sides.append((None, res, list(seen), list(trace), len(draws) - len(pool), sampler.adjust_steps(steps_n) if hasattr(sampler, "adjust_steps") else None))
        except Exception as err:  # noqa: BLE001
            sides.append((err, None, None, None, None, None))
    (re_, rres, rseen, rtrace, rused, radj), (pe, pres, pseen, ptrace, pused, padj) = sides
    if re_ or pe:
        return None if type(re_) is type(pe) else f"reference {re_!r}, here {pe!r}"
    if kind in ("bf16", "f16"):  # 16-bit tensors: the reference computes in the tensor dtype, one rounded op at a time -- bit for bit (adaptive RKMoire and the SPC adapter aside: fused blends / error norms steer them)
        if type(rres) is not type(pres) or rres.dtype != pres.dtype:
            return f"result is a {type(pres).__name__} of {getattr(pres, 'dtype', None)}, the reference's of {getattr(rres, 'dtype', None)}"
        if radj != padj or rused != pused or len(rseen) != len(pseen):
            return f"adjust_steps {radj}/{padj}, draws {rused}/{pused}, model calls {len(rseen)}/{len(pseen)}"
        if not torch.isfinite(rres.float()).all():
            return None
        if "RKMoire" in text or "SPC" in text:
            err = ((pres.double() - rres.double()).abs().max() / rres.double().abs().max().clamp_min(1e-30)).item()
            return None if err <= 0.05 else f"result differs: rel inf-norm {err:.3g}"
        return None if torch.equal(rres, pres) else f"{(rres != pres).sum().item()} elements differ from the reference's bits (max {(rres.double() - pres.double()).abs().max().item():.3g})"
    if type(rres) is not type(pres) or np.asarray(rres).dtype != np.asarray(pres).dtype:
        return f"result is a {type(pres).__name__} of {np.asarray(pres).dtype}, the reference's a {type(rres).__name__} of {np.asarray(rres).dtype}"
    rres, pres = torch.as_tensor(np.asarray(rres, dtype=np.float64)), torch.as_tensor(np.asarray(pres, dtype=np.float64))
    if not torch.isfinite(rres).all():
        return None if not torch.isfinite(pres).all() else "reference non-finite, here finite"
    if radj != padj or rused != pused or len(rseen) != len(pseen) or len(rtrace) != len(ptrace):
        return f"adjust_steps {radj}/{padj}, draws {rused}/{pused}, model calls {len(rseen)}/{len(pseen)}, callbacks {len(rtrace)}/{len(ptrace)}"
    close = lambda a, b: np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-5 if kind == "f32" else 1e-9, atol=1e-5 if kind == "f32" else 1e-9)  # noqa: E731
    if rseen and not close(rseen, pseen):
        return "the (t, sigma, alpha) the model was called at differ"
    if rtrace and not close(rtrace, ptrace):
        return "callback traces differ"
    err = ((pres - rres).abs().max() / rres.abs().max().clamp_min(1e-30)).item()
    return None if err <= (5e-5 if kind == "f32" else 1e-9) else f"result differs: rel inf-norm {err:.3g}"  # (float32: a free-running chain of up to nine steps of <= 1e-5 each)


# ---- structured noise on host tensors: the reference's generators vs pytorch/host_noise.py, same CPU generator seeds -> same bits -------
def _noise_spec(rng):
    rank = rng.choice((2, 3, 3, 3, 4))
    unit = tuple(rng.choice((1, 2, 3, 4, 6, 8, 9, 12, 16, 17, 24, 32)) for _ in range(rank))
    kind = rng.choice(("Random", "Offset", "Pyramid", "Colored"))
    if kind == "Offset":
        dims = tuple(sorted(rng.sample(range(rank), rng.randint(1, rank))))
        props = rng.choice(("None", f"N.OffsetProps(dims={dims}, strength={rng.choice((0.2, 0.5, 1.3))}, static={rng.choice((True, False))})"))
    elif kind == "Pyramid":
        dims = tuple(rng.sample(range(-rank, rank), rng.randint(1, min(rank, 3))))
        props = rng.choice(("None", f"N.PyramidProps(dims={dims}, strength={rng.choice((0.3, 0.6, 1.0))}, depth={rng.choice((1, 2, 3, 99))}, static={rng.choice((True, False))})"))
    elif kind == "Colored":
        props = rng.choice(("None", f"N.ColoredProps(energy={rng.choice((None, 2.5, 0.5))}, color_start={rng.choice((0.25, 1.5, 0, -1))}, color_end={rng.choice((-2, -3, 0, 1))}, color_curve={rng.choice((2, 0, 1, 0.5))})"))
    else:
        props = "None"
    step = rng.choice(("None", "Step(0.0, 0.05)", "Step(0.45, 0.5)", "Step(0.95, 1.0)", "Step(0.3, 0.9)"))
    dtype = rng.choice(("float32", "float32", "float64", "bfloat16"))
    return kind, unit, props, step, dtype, rng.randint(1, 3)


def one_noise(seed: int) -> str | None:
    from skrample.common import Step as RStep

    from skrample_amd.common import Step as OStep
    from skrample_amd.pytorch import noise as ON

    kind, unit, props, step, dtype, draws = _noise_spec(random.Random(seed))
    dt = getattr(torch, dtype)
    from skrample_amd.pytorch import host_noise as HN

    sides = []
    for N, StepT in ((MG.RN, RStep), (ON, OStep)):
        env = {"N": N, "Step": StepT}
        gens = [torch.Generator().manual_seed(seed + k) for k in range(2)]
        try:
            pr = eval(props, env)
            if N is MG.RN:
                gen = N.BatchTensorNoise.from_batch_inputs(getattr(N, kind), unit, gens, *(() if pr is None else (pr,)), dtype=dt) if pr is not None else N.BatchTensorNoise.from_batch_inputs(getattr(N, kind), unit, gens, dtype=dt)
            else:  # (what the wrapper builds for host-resident latents, diffusers.py::_make_noise_generator)
                gen = ON.HostRandomBatch(unit, gens, dt) if kind == "Random" else HN.HostStructuredBatch(getattr(N, kind), unit, gens, pr, dt)
            outs = [torch.as_tensor(gen.generate(eval(step, env))) for _ in range(draws)]
            sides.append((None, outs))
        except Exception as err:  # noqa: BLE001
            sides.append((err, None))
    (re_, r), (pe, p) = sides
    if re_ or pe:
        return None if type(re_) is type(pe) else f"reference {re_!r}, here {pe!r}"
    for i, (a, b) in enumerate(zip(p, r)):
        if a.dtype != b.dtype or a.shape != b.shape:
            return f"draw {i}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}"
        if not torch.equal(a.cpu(), b) and not (torch.isnan(a.cpu()) == torch.isnan(b)).all():
            return f"draw {i}: NaN pattern differs"
        same = torch.equal(torch.nan_to_num(a.cpu().double(), nan=7.0), torch.nan_to_num(b.double(), nan=7.0))
        if not same:
            d = (a.cpu().double() - b.double()).abs().max().item()
            return f"draw {i}: bits differ (max diff {d:.3g})"
    return None


# ---- diffusers configs: from_diffusers_config over random config dicts, then set_timesteps with random arguments ----------------------------------
def _config_spec(rng):
    cfg: dict = {}
    if rng.random() < 0.85:
        cfg["_class_name"] = rng.choice(
            ("DDIMScheduler", "DDPMScheduler", "DPMSolverMultistepScheduler", "DPMSolverSDEScheduler", "EulerAncestralDiscreteScheduler", "EulerDiscreteScheduler",
             "FlowMatchEulerDiscreteScheduler", "IPNDMScheduler", "MiniMaxH3Scheduler", "UniPCMultistepScheduler", "PNDMScheduler")
        )  # fmt: skip
    for key, values in (
        ("prediction_type", ("epsilon", "flow", "sample", "v_prediction", "other")),
        ("beta_schedule", ("linear", "scaled_linear", "squaredcos_cap_v2")),
        ("algorithm_type", ("dpmsolver", "dpmsolver++", "sde-dpmsolver", "sde-dpmsolver++")),
        ("use_flow_sigmas", (True, False)),
        ("use_beta_sigmas", (True, False)),
        ("use_exponential_sigmas", (True, False)),
        ("use_karras_sigmas", (True, False)),
        ("rescale_betas_zero_snr", (True, False)),
        ("shift", (1.0, 3.0, 1.7)),
        ("flow_shift", (2.0, 5.0)),
        ("solver_order", (1, 2, 3)),
        ("num_train_timesteps", (1000, 500)),
        ("beta_start", (0.00085, 0.0001)),
        ("beta_end", (0.012, 0.02)),
        ("timestep_spacing", ("leading", "trailing")),
    ):
        if rng.random() < 0.3:
            cfg[key] = rng.choice(values)
    extra = rng.choice(("", "", "", ", sampler=T.Adams", ", schedule=S.Linear", ", schedule=S.ZSNR", ", subschedule=S.Karras", ", subschedule=S.Beta, subschedule_props={'alpha': 0.8}",
                        ", schedule_modifiers=[(S.Hyper, {})]", ", schedule_modifiers=[(S.FlowShift, {'shift': 2.0})], modifier_merge_strategy=W.MergeStrategy.Ours", ", model=M.VelocityModel()",
                        ", sampler_props={'order': 2}", ", schedule_props={'base_timesteps': 800}", ", invert_prediction=True", ", allow_dynamic=False"))  # fmt: skip
    kind = rng.choice(("W.SkrampleWrapperScheduler", "W.SkrampleWrapperScheduler", "W.RKUltraWrapperScheduler", "W.DynasauRKWrapperScheduler"))
    if kind != "W.SkrampleWrapperScheduler" and ("sampler" in extra):
        extra = ""
    call = rng.choice(("set_timesteps({n})", "set_timesteps({n})", "set_timesteps({n}, mu=0.8)", "set_timesteps(num_inference_steps={n}, device='cpu')", "set_timesteps(timesteps=[900.0, 500.0, 100.0])", "set_timesteps(sigmas=[1.0, 0.5, 0.25, 0.1])"))
    return cfg, kind, extra, call.format(n=rng.randint(1, 12))


def one_config(seed: int) -> str | None:
    cfg, kind, extra, call = _config_spec(random.Random(seed))
    sides = []
    for names in (REF, OWN):
        try:
            w = eval(f"{kind}.from_diffusers_config(cfg{extra})", {**names, "cfg": dict(cfg)})
            made = (repr(getattr(w, "sampler", None)), repr(w.schedule), repr(w.model), w.invert_prediction, sorted((k, repr(v)) for k, v in w.config.items()))
            eval(f"w.{call}", {"w": w})
            after = (repr(w.schedule), w.timesteps.tolist(), np.asarray(w.schedule_np).tolist(), sorted((k, repr(v)) for k, v in w.config.items()), w.order, w.init_noise_sigma if hasattr(w, "init_noise_sigma") else None)
            sides.append((None, made, after))
        except Exception as err:  # noqa: BLE001
            sides.append((err, None, None))
    (re_, rmade, rafter), (pe, pmade, pafter) = sides
    if re_ or pe:
        return None if type(re_) is type(pe) else f"reference {re_!r}, here {pe!r}"
    strip = lambda text: text.replace("skrample_amd.", "skrample.")  # noqa: E731
    if strip(repr(rmade)) != strip(repr(pmade)):
        return f"constructed objects differ:\n      {rmade}\n      {pmade}"
    if strip(repr(rafter[0])) != strip(repr(pafter[0])) or rafter[4:] != pafter[4:] or strip(repr(rafter[3])) != strip(repr(pafter[3])):
        return f"after {call}: {rafter[0]} order {rafter[4:]} vs {pafter[0]} order {pafter[4:]} / config {rafter[3]} vs {pafter[3]}"
    if not np.allclose(rafter[1], pafter[1], rtol=0, atol=1e-9, equal_nan=True) or not np.allclose(rafter[2], pafter[2], rtol=1e-12, atol=1e-12, equal_nan=True):
        return f"after {call}: timesteps / points differ"
    return None


# ---- the sampler-level API on host tensors (no wrapper): sampler.sample(...) free-running with its own history; 16-bit results bit for bit -------
def _sampler_spec(rng):
    text, dtype, shape, steps_n = MG._sweep_spec(rng)
    while not text.startswith("W.SkrampleWrapperScheduler("):
        text, dtype, shape, steps_n = MG._sweep_spec(rng)
    inner = text[len("W.SkrampleWrapperScheduler(") : -1]
    for opt in (", invert_prediction=True", ", compute_scale=torch.float64"):
        inner = inner.replace(opt, "")
    return f"({inner})", dtype, shape, steps_n


def one_sampler(seed: int) -> str | None:
    from skrample.common import Step as RStep

    from skrample_amd.common import Step as OStep

    text, dtype, shape, steps_n = _sampler_spec(random.Random(seed))
    dt = getattr(torch, dtype)
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(shape, generator=g).to(dt)
    outs = [torch.randn(shape, generator=g).to(dt) for _ in range(steps_n)]
    nzs = [torch.randn(shape, generator=g).to(dt) for _ in range(steps_n)]
    sides = []
    fused16 = dt in (torch.bfloat16, torch.float16) and "T.SPC(" in text and "power=1," not in text  # (the signed-power blend stays fused: not the reference's 16-bit chain)
    for names, StepT in ((REF, RStep), (OWN, OStep)):
        try:
            sampler, schedule, model = eval(text, names)
            x, prev, got = x0, [], []
            for i in range(steps_n):
                rec = sampler.sample(x, outs[i], StepT.from_int(i, steps_n), model, schedule, nzs[i] if sampler.require_noise else None, tuple(prev))
                final, pred = torch.as_tensor(rec.final), rec.prediction
                pred = torch.as_tensor(pred.materialize() if hasattr(pred, "materialize") else pred)
                got.append((final, pred))
                prev.append(rec)
                x = final
            sides.append((None, got))
        except Exception as err:  # noqa: BLE001
            sides.append((err, None))
    (re_, r), (pe, p) = sides
    def against_float64(slack: float):
        # Arithmetic that differs by design (fp32 accumulation, one rounding, against the reference's chain of 16-bit ops): free-running, the reference's chain
        # drifts from its own float64 run by several per cent within a few steps (seed 720187: 5.8 % at step 4, this engine 0.7 %).  The bar: at every step
        # this engine is as close to the reference's float64 run as the reference's own run in `dtype` is (x 1.5 + slack).
        sampler, schedule, model = eval(text, REF)
        x, prev = x0.double(), []
        for i in range(steps_n):
            rec = sampler.sample(x, outs[i].double(), RStep.from_int(i, steps_n), model, schedule, nzs[i].double() if sampler.require_noise else None, tuple(prev))
            exact = torch.as_tensor(rec.final).double()
            prev.append(rec)
            x = exact
            if not torch.isfinite(exact).all() or not torch.isfinite(r[i][0].float()).all():
                return None
            scale = exact.abs().max().clamp_min(1e-30)
            theirs, ours = ((r[i][0].double() - exact).abs().max() / scale).item(), ((p[i][0].double() - exact).abs().max() / scale).item()
            if p[i][0].dtype != r[i][0].dtype or ours > 1.5 * theirs + slack:
                return f"step {i} final: {ours:.3g} from the reference's float64 run, the reference's own {dtype} run {theirs:.3g}"
        return None

    if fused16 and re_ is None and pe is None:
        return against_float64(0.005)
    if re_ or pe:
        if re_ is None and isinstance(pe, ZeroDivisionError):
            return None if not all(torch.isfinite(a.float()).all() and torch.isfinite(b.float()).all() for a, b in r) else f"here {pe!r}, reference finite"
        if re_ is not None and isinstance(pe, ZeroDivisionError):
            return None  # (the same documented class one step on: the reference's tensors went inf / nan where this side raised, and a later step of its run failed on them)
        return None if type(re_) is type(pe) else f"reference {re_!r}, here {pe!r}"
    exact = dt in (torch.bfloat16, torch.float16) and ("T.SPC(" not in text or "power=1," in text)  # (the signed-power blend stays fused)
    for i, ((fa, pa), (fb, pb)) in enumerate(zip(p, r)):
        for name, a, b in (("final", fa, fb), ("prediction", pa, pb)):
            if a.dtype != b.dtype or a.shape != b.shape:
                return f"step {i} {name}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}"
            if not torch.isfinite(b.float()).all():
                return None
            if exact:
                if not torch.equal(a, b):
                    return f"step {i} {name}: {(a != b).sum().item()} elements differ from the reference's bits (max {(a.double() - b.double()).abs().max().item():.3g})"
            else:
                scale = b.double().abs().max().clamp_min(1e-30)
                err = ((a.double() - b.double()).abs().max() / scale).item()
                bar = 1e-5 if dt in (torch.float32, torch.float64) else 0.03  # (SPC's signed-power blend on 16-bit: fused form vs the reference's 16-bit chain, one step of it)
                if err > bar:
                    if dt == torch.float32 and err < 1e-4:
                        # a free-running fp32 chain a hair over the bar (ill-conditioned last steps: 1.2e-5 ... 1.5e-5 in 2 of 2500 configurations): judged like the
                        # 16-bit class above -- no farther from the reference's float64 run than the reference's own fp32 run
                        return against_float64(1e-6)
                    return f"step {i} {name}: rel inf-norm {err:.3g}"
    return None


# ---- schedules: random compositions and parameters, every public evaluation ------------------------------------------------------------------------
def _schedule_spec(rng):
    num = lambda lo, hi: repr(round(rng.uniform(lo, hi), 4))  # noqa: E731
    if rng.random() < 0.45:
        base = rng.choice(("S.Linear()", f"S.Linear(sigma_start={num(0.5, 20)})", f"S.Linear(base_timesteps={rng.choice((1000, 1, -1000, 500))})", "S.Linear(custom_space=S.VariancePreserving())"))
    else:
        base = rng.choice(("S.Scaled()", "S.ZSNR()", f"S.Scaled(beta_start={num(0.0001, 0.001)}, beta_end={num(0.01, 0.03)}, beta_scale={rng.choice((1, 2, 3))})",
                           f"S.Scaled(base_timesteps={rng.choice((1000, -1000, 250))})", f"S.ZSNR(beta_scale={rng.choice((1, 2))})"))  # fmt: skip
    text = base
    if rng.random() < 0.5:
        text = rng.choice((f"S.Karras({text}, rho={num(1, 9)}, steps={rng.randint(2, 40)})", f"S.Exponential({text}, rho={num(0.5, 3)}, steps={rng.randint(2, 40)})",
                           f"S.Beta({text}, alpha={num(0.3, 1.5)}, beta={num(0.3, 1.5)})", f"S.Probit({text}, scale={num(1, 5)})", f"S.Karras({text})", f"S.Beta({text})"))  # fmt: skip
    for _ in range(rng.choice((0, 0, 1, 1, 2))):
        text = rng.choice((f"S.FlowShift({text}, shift={num(0.5, 6)})", f"S.Hyper({text}, scale={num(-3, 4)}, tail={rng.choice((True, False))})",
                           f"S.Sinner({text}, count={num(-4, 4)}, scale={num(0.5, 3)})", f"S.Hyper({text})", f"S.Sinner({text})"))  # fmt: skip
    return text, rng.randint(1, 30), [round(rng.random(), 6) for _ in range(5)] + [0.0, 1.0]


def one_schedule(seed: int) -> str | None:
    text, steps_n, ts = _schedule_spec(random.Random(seed))
    sides = []
    for names in (REF, OWN):
        got = {}
        try:
            sch = eval(text, names)
        except Exception as err:  # noqa: BLE001
            sides.append({"construct": err})
            continue
        for name, fn in (
            ("schedule_np", lambda: np.asarray(sch.schedule_np(steps_n))), ("points", lambda: np.asarray(sch.points(ts))), ("ipoints", lambda: np.asarray(sch.ipoints(ts))),
            ("ipoint", lambda: np.asarray([sch.ipoint(t) for t in ts])), ("point", lambda: np.asarray([sch.point(t) for t in ts])), ("ends", lambda: np.asarray([sch.point_0, sch.point_1])),
            ("space", lambda: np.asarray([sch.space.regularize(0.37), sch.space.alpha(0.37), *sch.space.regularize(np.asarray([0.1, 2.0]))], dtype=np.float64)),
            ("schedule", lambda: np.asarray(sch.schedule(steps_n))), ("sigmas", lambda: np.asarray(sch.sigmas(steps_n))), ("timesteps", lambda: np.asarray(sch.timesteps(steps_n))),
            ("repr", lambda: np.asarray([0.0]) if repr(sch) else None),
        ):  # fmt: skip
            try:
                got[name] = fn()
            except Exception as err:  # noqa: BLE001
                got[name] = err
        sides.append(got)
    r, p = sides
    for name in r:
        a, b = p.get(name), r[name]
        if isinstance(a, Exception) or isinstance(b, Exception):
            if type(a) is not type(b):
                return f"{name}: reference {b!r}, here {a!r}"
            continue
        if a is None or a.shape != b.shape or not np.allclose(a, b, rtol=1e-10, atol=1e-10, equal_nan=True):
            return f"{name}: differs\n      {b.tolist()}\n      {None if a is None else a.tolist()}"
    return None


# ---- the wrapper's own noise: generator=None (seeds from the sample), one generator, a list, a list of the wrong length; every noise type; host tensors ----
def _generator_spec(rng):
    batch = rng.choice((1, 1, 2, 3))
    shape = (batch, rng.randint(1, 4), rng.choice((4, 8)), rng.choice((5, 8)))
    noise = rng.choice(("Random", "Random", "Offset", "Pyramid", "Colored"))
    if rng.random() < 0.7:
        text = f"W.SkrampleWrapperScheduler({rng.choice(('T.DPM(order=2, stochasticity=1)', 'T.Euler(stochasticity=0.5)', 'T.UniPC(order=2, stochasticity=1)'))}, S.Scaled(), noise_type=N.{noise})"
    else:
        text = f"W.RKUltraWrapperScheduler(S.Scaled(), sampler_order={rng.randint(1, 4)}, stochasticity=1, noise_type=N.{noise})"
    return text, shape, rng.choice(("none", "single", "list", "shortlist")), rng.choice(("float32", "bfloat16"))


def one_generator(seed: int) -> str | None:
    from skrample_amd.pytorch import noise as ON

    text, shape, mode, dtype = _generator_spec(random.Random(seed))
    dt = getattr(torch, dtype)
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(shape, generator=g).to(dt)
    outs = [torch.randn(shape, generator=g).to(dt) for _ in range(16)]
    def gens():
        return {"none": None, "single": torch.Generator().manual_seed(5), "list": [torch.Generator().manual_seed(7 + i) for i in range(shape[0])], "shortlist": [torch.Generator().manual_seed(9)]}[mode]

    # Teacher-forced on the reference's state, like the other modes: a free-running bf16 loop turns one last-place difference of step i
    # (fp32 registers rounded once, against the reference's rounded fp32 chain) into several of them at step i + 1 (seeds 3001965, 3002873).
    made = []
    for names, N in ((REF, MG.RN), (OWN, ON)):
        try:
            w = eval(text, {**names, "N": N})
            w.set_timesteps(3)
            made.append((w, gens()))
        except Exception as err:  # noqa: BLE001
            made.append(err)
    if any(isinstance(m, Exception) for m in made):
        return None if type(made[0]) is type(made[1]) else f"reference {made[0]!r}, here {made[1]!r}"
    (wr, gr), (wp, gp) = made
    x = x0
    for i, t in enumerate(wr.timesteps):
        got = []
        for w, gen in ((wr, gr), (wp, gp)):
            try:
                got.append(torch.as_tensor(w.step(outs[i], t, x, generator=gen, return_dict=False)[0]))
            except Exception as err:  # noqa: BLE001
                got.append(err)
        b, a = got
        if isinstance(a, Exception) or isinstance(b, Exception):
            return None if type(a) is type(b) else f"step {i}: reference {b!r}, here {a!r}"
        if a.dtype != b.dtype or not torch.allclose(a.double(), b.double(), rtol=1e-5 if dt == torch.float32 else 2.0**-7, atol=1e-5):
            return f"step {i}: max diff {(a.double() - b.double()).abs().max().item():.3g} (the draws differ, or the step)"
        x = b
    return None


# ---- wrappers without a compute scale on 16-bit host tensors: the reference computes in the tensor dtype, op by op -- results bit for bit ------------------
def _native_spec(rng):
    from sweep_grammar import native_spec

    return native_spec(rng)


def one_native(seed: int) -> str | None:
    text, dtype, shape, steps_n = _native_spec(random.Random(seed))
    dt = getattr(torch, dtype)
    g = torch.Generator().manual_seed(seed)
    made = []
    for names in (REF, OWN):
        try:
            w = eval(text, names)
            w.set_timesteps(steps_n)
            len(w.timesteps)  # (the Runge-Kutta wrappers build their table here, and refuse some schedules with an assertion)
            made.append(w)
        except Exception as err:  # noqa: BLE001
            made.append(err)
    if any(isinstance(m, Exception) for m in made):
        return None if type(made[0]) is type(made[1]) else f"set_timesteps: reference {made[0]!r}, here {made[1]!r}"
    r, p = made
    if not torch.isfinite(r.timesteps).all():
        return None
    x = torch.randn(shape, generator=g).to(dt)
    n = len(r.timesteps)
    outs = [torch.randn(shape, generator=g).to(dt) for _ in range(n)]
    noises = [torch.randn(shape, generator=g).to(dt) for _ in range(n)]  # (get_step_noise hands the draw over in the sample's dtype: diffusers.py:346)
    r._noise_generator, p._noise_generator = MG._Injected(list(noises)), _Replay(list(noises))
    for i, (tr, tp) in enumerate(zip(r.timesteps, p.timesteps)):
        sides = []
        for w, t in ((r, tr), (p, tp)):
            try:
                got = w.step(outs[i], t, x, return_dict=False)
                sides.append([torch.as_tensor(v.materialize() if isinstance(v, lazy.LazyTensor) else v) for v in got])
            except Exception as err:  # noqa: BLE001
                sides.append(err)
        a, b = sides
        if isinstance(a, Exception) or isinstance(b, Exception):
            if isinstance(a, Exception) and isinstance(b, Exception):
                return None
            if isinstance(b, Exception) and not all(torch.isfinite(v.float()).all() for v in a):
                return None
            return f"step {i}: reference {a if isinstance(a, Exception) else 'ok'!r}, here {b if isinstance(b, Exception) else 'ok'!r}"
        if not all(torch.isfinite(v.float()).all() for v in a):
            return None
        for name, u, v in zip(("prev", "pred"), a, b):
            if u.dtype != v.dtype or not torch.equal(u, v):
                return f"step {i} {name}: {(u != v).sum().item()} elements differ from the reference's bits (max {(u.double() - v.double()).abs().max().item():.3g})"
        x = a[0]
    return None


# ---- the sampler-level API on what else the reference's generic `T` covers: Python floats and numpy arrays (float32 / float64) -----------------------------
def _array_spec(rng):
    text, _dtype, shape, steps_n = _sampler_spec(rng)
    return text, rng.choice(("float", "f64", "f32", "f64")), shape, steps_n


def one_array(seed: int) -> str | None:
    from skrample.common import Step as RStep

    from skrample_amd.common import Step as OStep

    text, kind, shape, steps_n = _array_spec(random.Random(seed))
    g = np.random.default_rng(seed)
    mk = (lambda: float(g.standard_normal())) if kind == "float" else (lambda: g.standard_normal(shape).astype(np.float64 if kind == "f64" else np.float32))
    x0, outs, nzs = mk(), [mk() for _ in range(steps_n)], [mk() for _ in range(steps_n)]
    def run(names, StepT, cast=lambda v: v):
        try:
            sampler, schedule, model = eval(text, names)
            x, prev, got = cast(x0), [], []
            for i in range(steps_n):
                rec = sampler.sample(x, cast(outs[i]), StepT.from_int(i, steps_n), model, schedule, cast(nzs[i]) if sampler.require_noise else None, tuple(prev))
                pred = rec.prediction.materialize() if hasattr(rec.prediction, "materialize") else rec.prediction
                got.append((rec.final, pred))
                prev.append(rec)
                x = rec.final
            return got
        except Exception as err:  # noqa: BLE001
            return err

    r, p = run(REF, RStep), run(OWN, OStep)
    yardstick = None  # the reference's own run on the same inputs widened to float64, made when a float32 comparison fails
    if isinstance(r, Exception) or isinstance(p, Exception):
        if isinstance(r, Exception) and isinstance(p, Exception):
            return None
        if isinstance(p, ZeroDivisionError) and not all(np.isfinite(np.asarray(v)).all() for pair in r for v in pair):
            return None  # (singular point: the reference's arrays fill with inf / nan, the engine raises)
        return f"reference {r if isinstance(r, Exception) else 'ok'!r}, here {p if isinstance(p, Exception) else 'ok'!r}"
    for i, ((fa, pa), (fb, pb)) in enumerate(zip(p, r)):
        if type(fa) is not type(fb):
            return f"step {i}: result is a {type(fa).__name__}, the reference's a {type(fb).__name__}"
        for name, a, b in (("final", fa, fb), ("prediction", pa, pb)):
            a, b = np.asarray(a), np.asarray(b)
            if a.shape != b.shape or a.dtype != b.dtype:
                return f"step {i} {name}: {a.dtype}{a.shape} vs {b.dtype}{b.shape}"
            if not np.isfinite(b).all():
                return None
            tol = 1e-5 if kind == "f32" else 1e-9  # (float32 inputs: float32 arithmetic somewhere upstream even where numpy promoted the result)
            if not np.allclose(a, b, rtol=tol, atol=tol * max(1.0, float(np.abs(b).max()))):
                # A free-running float32 loop of a high-order solver where the result is a small difference of large terms (seed 3000401: UniPC-6,
                # prediction 5.5 out of operands near 90): the reference's float32 run is itself 1.5e-4 from its float64 run there.  Ours
                # may be as far from that float64 run as the reference's own float32 run is (and a little more), no further.
                if kind == "f32":
                    yardstick = yardstick or run(REF, RStep, lambda v: v.astype(np.float64))
                    if not isinstance(yardstick, Exception):
                        y = np.asarray(yardstick[i][0 if name == "final" else 1])
                        if float(np.abs(a - y).max()) <= 1.5 * float(np.abs(b - y).max()) + tol * max(1.0, float(np.abs(b).max())):
                            continue
                return f"step {i} {name}: max diff {float(np.abs(a - b).max()):.3g}"
    return None


# ---- the model transforms called directly (to_x / from_x / forward / backward / ModelConvert) over every operand kind; 16-bit tensors bit for bit ---------
MODEL_TEXTS = ("M.DataModel()", "M.NoiseModel()", "M.FlowModel()", "M.VelocityModel()", "M.ScaleX()", "M.ScaleX(bias=-1.5)")


def _model_spec(rng):
    sig0 = rng.uniform(0.05, 0.95)
    sig1 = rng.uniform(0.02, sig0)
    flow = rng.random() < 0.5
    points = ((500.0, sig0, 1 - sig0), (300.0, sig1, 1 - sig1)) if flow else ((500.0, sig0, math.sqrt(1 - sig0**2)), (300.0, sig1, math.sqrt(1 - sig1**2)))
    return (rng.choice(MODEL_TEXTS), rng.choice(MODEL_TEXTS), rng.choice(("float", "f64arr", "f32arr", "t64", "t32", "tbf16", "tf16")),
            rng.choice(("to_x", "from_x", "forward", "forward_noise", "backward", "backward_noise", "output_to", "output_from", "wrap")), points, rng.choice((0, 0.5, 1)))  # fmt: skip


def one_model(seed: int) -> str | None:
    import skrample.sampling.models as RM
    from skrample.common import DeltaPoint as RDelta
    from skrample.common import Point as RPoint

    import skrample_amd.sampling.models as OM
    from skrample_amd.common import DeltaPoint as ODelta
    from skrample_amd.common import Point as OPoint

    m1, m2, kind, op, points, eta = _model_spec(random.Random(seed))
    g = torch.Generator().manual_seed(seed)

    def make():
        t = torch.randn([2, 3], generator=g, dtype=torch.float64)
        return {"float": lambda: float(t[0, 0]), "f64arr": t.numpy, "f32arr": lambda: t.float().numpy(), "t64": lambda: t, "t32": t.float, "tbf16": t.bfloat16, "tf16": t.half}[kind]()

    s_, o_, n_ = make(), make(), make()
    sides = []
    for names, P, D, Mod in ((REF, RPoint, RDelta, RM), (OWN, OPoint, ODelta, OM)):
        try:
            a, b, p0, p1 = eval(m1, names), eval(m2, names), P(*points[0]), P(*points[1])
            d = D(p0, p1)
            v = {
                "to_x": lambda: a.to_x(s_, o_, p0), "from_x": lambda: a.from_x(s_, o_, p0), "forward": lambda: a.forward(s_, o_, d), "forward_noise": lambda: a.forward(s_, o_, d, n_, eta),
                "backward": lambda: a.backward(s_, o_, d), "backward_noise": lambda: a.backward(s_, o_, d, n_, eta), "output_to": lambda: Mod.ModelConvert(a, b).output_to(s_, o_, p0),
                "output_from": lambda: Mod.ModelConvert(a, b).output_from(s_, o_, p0), "wrap": lambda: Mod.ModelConvert(a, b).wrap_model_call(lambda x, t, sg, al: x * 0.5)(s_, *p0),
            }[op]()  # fmt: skip
            sides.append(v.materialize() if isinstance(v, lazy.LazyTensor) else v)
        except Exception as err:  # noqa: BLE001
            sides.append(err)
    r, p = sides
    if isinstance(r, Exception) or isinstance(p, Exception):
        return None if type(r) is type(p) else f"reference {r!r}, here {p!r}"
    if type(r) is not type(p) or getattr(r, "dtype", None) != getattr(p, "dtype", None):
        return f"a {type(p).__name__} of {getattr(p, 'dtype', None)}, the reference's a {type(r).__name__} of {getattr(r, 'dtype', None)}"
    if kind in ("tbf16", "tf16"):
        ok = torch.equal(torch.isnan(r), torch.isnan(p)) and torch.equal(torch.nan_to_num(r), torch.nan_to_num(p))
        return None if ok else f"{(r != p).sum().item()} elements differ from the reference's bits (max {(r.double() - p.double()).abs().max().item():.3g})"
    ra, pa = np.asarray(r, dtype=np.float64), np.asarray(p, dtype=np.float64)
    if not np.isfinite(ra).all():
        return None
    tol = 1e-5 if kind in ("f32arr", "t32") else 1e-12
    return None if np.allclose(pa, ra, rtol=tol, atol=tol * max(1.0, float(np.abs(ra).max()))) else f"max diff {float(np.abs(pa - ra).max()):.3g}"


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    found = 0
    which = sys.argv[3] if len(sys.argv) > 3 else "wrapper"
    for seed in range(first, first + count):
        spec, run = {"wrapper": (MG._sweep_spec, one), "functional": (_functional_spec, one_functional), "noise": (_noise_spec, one_noise), "config": (_config_spec, one_config), "sampler": (_sampler_spec, one_sampler), "schedule": (_schedule_spec, one_schedule), "generator": (_generator_spec, one_generator), "native": (_native_spec, one_native), "array": (_array_spec, one_array), "model": (_model_spec, one_model)}[which]
        text = spec(random.Random(seed))
        try:
            why = run(seed)
        except Exception as err:  # noqa: BLE001
            why = f"harness error {err!r}"
        if why:
            found += 1
            print(f"seed {seed}: {text}\n    {why}", flush=True)
    print(f"{count} configurations, {found} differences")
