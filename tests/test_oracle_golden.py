"""Pin the oracle (oracle/skr_oracle) to the reference: its committed known answers and the
fixtures captured by importing the reference itself (tools/make_golden.py).  CPU only."""

import math
import random

import numpy as np
import pytest
import torch
from cases import MODELS, NATIVE16_ORACLE, NATIVE16_TAGS, SCHEDULES, from_bits, native16_case, oracle_schedule
from conftest import eq_nan, load_npz

from skr_oracle import noise as ON
from skr_oracle import predictors as OP
from skr_oracle import rk as OK
from skr_oracle import samplers as OA
from skr_oracle import scalars as OC
from skr_oracle import schedules as OS
from skr_oracle import wrapper as OW

ORACLE_SCHEDULE = {"Linear": OS.linear, "Scaled": OS.scaled}
ORACLE_MODEL = {"DataModel": "data", "FlowModel": "flow", "VelocityModel": "v"}


def trajectory(sampler: str, sched, pred, steps=7, seed=42):
    out = []
    random.seed(seed)
    hs = OS.hyper(sched)
    model = lambda x, t, s, a: x - math.sin(t)  # noqa: E731
    rng = lambda _: random.random()  # noqa: E731
    cb = lambda x, i, d: out.append(x)  # noqa: E731
    if sampler == "RKUltra":
        loop = lambda x: OK.rk_loop(lambda st: OK.pick_tableau(2, {2: OK.TAB_HEUN}), x, model, pred, hs, steps, rng=rng, callback=cb)  # noqa: E731
    elif sampler == "DynasauRK":
        loop = lambda x: OK.rk_loop(lambda st: OK.dynasaur_tableau(st, 2), x, model, pred, hs, steps, rng=rng, callback=cb)  # noqa: E731
    else:
        cfg = OA.make(sampler.lower())
        loop = lambda x: OA.adapter_loop(cfg, x, model, pred, hs, steps, rng=rng, callback=cb)  # noqa: E731
    OA.generate(loop, None, rng, hs, steps)
    return out


def test_reference_sampler_goldens(kats):
    "reference tests/self_sampling.py:57-104 (tolerance there: 1e-3 %); the oracle is held to 1e-12 relative"
    assert len(kats["sampler_trajectories"]) == 24
    for key, ref in kats["sampler_trajectories"].items():
        s, sch, m = key.split("/")
        got = trajectory(s, ORACLE_SCHEDULE[sch](), ORACLE_MODEL[m], kats["measured_steps"], kats["measured_seed"])
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=0, err_msg=key)


def _oracle_schedule_by_label(label: str):
    # labels look like "Karras(Scaled())" / "Hyper(Hyper(Linear()))"
    env = {
        "Linear": OS.linear, "Scaled": OS.scaled, "Karras": OS.karras, "Beta": OS.beta, "Exponential": OS.exponential,
        "Probit": OS.probit, "Hyper": OS.hyper, "Sinner": OS.sinner,
    }  # fmt: skip
    return eval(label, {"__builtins__": {}}, env)  # noqa: S307 - labels come from our own fixture


def test_reference_schedule_goldens(kats):
    "reference tests/self_scheduling.py:27-46,94-100 (rtol 1e-5 there)"
    assert len(kats["schedule_points"]) == 14
    for label, ref in kats["schedule_points"].items():
        got = _oracle_schedule_by_label(label).points_np(np.linspace(1, 0, 7))
        np.testing.assert_allclose(got, np.asarray(ref), rtol=1e-5, err_msg=label)


def test_bashforth_kat(kats):
    "reference tests/miscellaneous.py:9-13"
    for n, coeffs in enumerate(kats["bashforth"]):
        assert np.allclose(coeffs, OC.bashforth(n + 1), atol=1e-12, rtol=1e-12)


def test_wrapper_tables(tables):
    for key, ref in tables["wrapper"].items():
        name, n = key.split("/")
        n = int(n)
        sched = oracle_schedule(name, n)
        d = OW.StepDriver(OA.make("euler"), sched)
        d.set_timesteps(n)
        assert eq_nan(d.timesteps.tolist(), ref["timesteps"]), key
        assert eq_nan(d.sigmas.tolist(), ref["sigmas"]), key
        assert eq_nan(d.table.tolist(), ref["schedule_np"]), key
        assert eq_nan(list(sched.point(0)), ref["point_0"]) and eq_nan(list(sched.ipoint(0.37)), ref["ipoint_0.37"])


def test_gamma_delta_zeta(tables):
    pred = {"data": "data", "eps": "eps", "flow": "flow", "v": "v", "scalex": ("scalex", 3)}
    for row in tables["gdz"]:
        sched = SCHEDULES[row["schedule"]][0]()
        p0, p1 = sched.ipoints(row["step"])
        try:
            got = [OP.gamma(pred[row["model"]], p0, p1, row["eta"]), OP.delta(pred[row["model"]], p0, p1, row["eta"]), OP.zeta(p0, p1, row["eta"])]
        except ZeroDivisionError:
            got = None
        ref = row["gdz"]
        if ref is None or got is None:
            assert ref is None and got is None, row
        elif isinstance(ref[0], str):
            assert [repr(v) for v in got] == ref, row
        else:
            assert got == ref, row


def test_effective_order(tables):
    kind = {"DPM": "dpm", "Adams": "adams", "UniP": "unip", "UniPC": "unipc"}
    for row in tables["effective_order"]:
        cfg = OA.make(kind[row["sampler"]], row["order"])
        got = [OA.effective_order(cfg, OC.stp_from_int(i, row["steps"]), row["n_previous"]) for i in range(row["steps"])]
        assert got == row["eo"], row
        assert OA.require_previous(cfg) == row["require_previous"]


def test_tableaux_and_rk_points(tables):
    named = {"RK1.Euler": OK.TAB_EULER, "RK2.Mid": OK.tab_rk2(1 / 2), "RK2.EES5_MIN": OK.tab_ees25(1 / 10), "RKE2.Heun": OK.TAB_HEUN,
             "RK2.EES7_MIN": OK.tab_ees27(1 / 14 * (5 - 3 * math.sqrt(2))), "SSP.RK4_5": OK.TAB_SSPRK4_5, "RKE5.CashKarp": OK.TAB_CASHKARP}  # fmt: skip
    for name, tab in named.items():
        ref = tables["tableaux"][name]
        assert [c for c, _ in tab[0]] == ref["c"] and [list(a) for _, a in tab[0]] == ref["a"] and list(tab[1]) == ref["b"], name
    for key, ref in tables["rk_points"].items():
        kind, name, order, steps = key.split("/")
        if kind != "rku" or int(order) > 6:
            continue
        pts = OK.rk_all_points(OK.pick_tableau(int(order)), SCHEDULES[name][0](), int(steps))
        assert [list(p) for p in pts] == ref["all"], key


STEP_CASES = {
    "cfg1": (lambda n: OW.StepDriver(OA.make("euler"), OS.scaled(), "eps"), torch.float32),
    "cfg2": (lambda n: OW.StepDriver(OA.make("dpm", 2, eta=1), OS.karras(OS.scaled(), steps=n), "eps"), torch.bfloat16),
    "cfg3": (lambda n: OW.StepDriver(OA.make("unipc", 3, eta=1), OS.linear(), "flow"), torch.bfloat16),
    "cfg4": (lambda n: OW.StepDriver(OA.make("adams", 4), OS.zsnr(), "v"), torch.bfloat16),
}


@pytest.mark.parametrize("name", STEP_CASES)
def test_step_fixtures(name):
    "oracle == reference SkrampleWrapperScheduler.step, bit for bit, on the BASELINE configs (reduced shape)"
    fx = load_npz(f"steps_{name}.npz")
    mk, dt = STEP_CASES[name]
    n = len(fx["timesteps"])
    drv = mk(n)
    drv.set_timesteps(n)
    assert np.array_equal(drv.timesteps.numpy(), fx["timesteps"])
    x = from_bits(fx["x0"], dt)
    for i in range(n):
        noise = torch.from_numpy(fx["noises"][i]) if fx["noises"].size else None
        prev, pred = drv.step(from_bits(fx["outs"][i], dt), drv.timesteps[i], x, noise=noise)
        assert torch.equal(prev, from_bits(fx["prev"][i], dt)), (name, i)
        assert torch.equal(pred, from_bits(fx["pred"][i], dt)), (name, i)
        x = prev


HIGH_ORDER_CASES = {  # tests/golden/steps_extra3.npz (round 3): Adams-Bashforth up to 9, UniP / UniPC beyond 3
    "unipc6_sde_eps": (lambda n: OW.StepDriver(OA.make("unipc", 6, eta=1), OS.scaled(), "eps"), torch.bfloat16),
    "adams9_eps_karras": (lambda n: OW.StepDriver(OA.make("adams", 9), OS.karras(OS.scaled(), steps=n), "eps"), torch.bfloat16),
    "adams6_v_zsnr": (lambda n: OW.StepDriver(OA.make("adams", 6), OS.zsnr(), "v"), torch.bfloat16),
    "unip7_flow": (lambda n: OW.StepDriver(OA.make("unip", 7), OS.linear(), "flow"), torch.float32),
    "unipc9_flow": (lambda n: OW.StepDriver(OA.make("unipc", 9), OS.linear(), "flow"), torch.bfloat16),
}


@pytest.mark.parametrize("name", HIGH_ORDER_CASES)
def test_high_order_step_fixtures(name):
    "oracle == reference SkrampleWrapperScheduler.step, bit for bit, for the high orders north_star names (12-step runs)"
    blob = load_npz("steps_extra3.npz")
    fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith(name + "/")}
    mk, dt = HIGH_ORDER_CASES[name]
    n = len(fx["timesteps"])
    drv = mk(n)
    drv.set_timesteps(n)
    assert np.array_equal(drv.timesteps.numpy(), fx["timesteps"])
    x = from_bits(fx["x0"], dt)
    for i in range(n):
        noise = torch.from_numpy(fx["noises"][i]) if fx["noises"].size else None
        prev, pred = drv.step(from_bits(fx["outs"][i], dt), drv.timesteps[i], x, noise=noise)
        assert torch.equal(prev, from_bits(fx["prev"][i], dt)), (name, i)
        assert torch.equal(pred, from_bits(fx["pred"][i], dt)), (name, i)
        x = prev


F64_CASES = {  # tests/golden/steps_extra4.npz (round 5): compute_scale=float64 over 16-bit latents
    "dpm2_f64_bf16": (lambda n: OW.StepDriver(OA.make("dpm", 2, eta=1), OS.scaled(), "eps", compute=torch.float64), torch.bfloat16),
    "unipc3_f64_f16": (lambda n: OW.StepDriver(OA.make("unipc", 3, eta=1), OS.linear(), "flow", compute=torch.float64), torch.float16),
    "adams4_f64_bf16": (lambda n: OW.StepDriver(OA.make("adams", 4), OS.zsnr(), "v", compute=torch.float64), torch.bfloat16),
}


@pytest.mark.parametrize("name", F64_CASES)
def test_float64_compute_scale_step_fixtures(name):
    "oracle == reference SkrampleWrapperScheduler.step with compute_scale=float64 on bf16 / fp16 tensors, bit for bit"
    blob = load_npz("steps_extra4.npz")
    fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith(name + "/")}
    mk, dt = F64_CASES[name]
    n = len(fx["timesteps"])
    drv = mk(n)
    drv.set_timesteps(n)
    assert np.array_equal(drv.timesteps.numpy(), fx["timesteps"])
    x = from_bits(fx["x0"], dt)
    for i in range(n):
        noise = torch.from_numpy(fx["noises"][i]) if fx["noises"].size else None
        prev, pred = drv.step(from_bits(fx["outs"][i], dt), drv.timesteps[i], x, noise=noise)
        assert torch.equal(prev, from_bits(fx["prev"][i], dt)), (name, i)
        assert torch.equal(pred, from_bits(fx["pred"][i], dt)), (name, i)
        x = prev


def test_float64_compute_scale_runge_kutta_fixture():
    "oracle RKDriver == reference RKUltraWrapperScheduler.step with compute_scale=float64 on bf16 tensors"
    blob = load_npz("steps_extra4.npz")
    fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith("rku4_f64_bf16/")}
    dt = torch.bfloat16
    drv = OW.RKDriver(OK.pick_tableau(4), OS.scaled(), "eps", "data", 0.5, compute=torch.float64)
    drv.set_timesteps(3)
    assert np.array_equal(drv.timesteps.numpy(), fx["timesteps"])
    x = from_bits(fx["x0"], dt)
    used = 0
    for i in range(len(fx["timesteps"])):
        def noise_fn(step=None):
            nonlocal used
            used += 1
            return torch.from_numpy(fx["noises"][used - 1])
        prev = drv.step(from_bits(fx["outs"][i], dt), drv.timesteps[i], x, noise_fn=noise_fn)
        assert torch.equal(prev, from_bits(fx["prev"][i], dt)), i
        x = prev
    assert used == int(fx["noise_used"])


def test_pyramid_dims_fixtures():
    """Pyramid over other `dims` subsets (reference noise.py:146-193): the oracle reproduces what the reference returned for the
    axis pairs it accepts; the single non-trailing axes it rejects (RuntimeError inside its own permute) are recorded as such"""
    fx = load_npz("noise_dims.npz")
    tags = sorted({k.rsplit("/", 1)[0] for k in fx if k.endswith("/out")})
    assert len(tags) == 6
    for tag in tags:
        unit = tuple(int(v) for v in tag.split("/")[1].split("x"))
        dims = tuple(int(d) for d in fx[f"{tag}/dims"])
        normals = [torch.from_numpy(fx[f"{tag}/normal{i}"]) for i in range(int(fx[f"{tag}/n_normals"]))]
        rp = ON.Replay(normals, fx[f"{tag}/uniforms"].tolist())
        got = ON.pyramid_noise(unit, rp.randn, rp.rand1, dims=dims)
        assert torch.equal(got, torch.from_numpy(fx[f"{tag}/out"])), tag
        assert not rp.normals
    rejected = sorted(k.split("/")[0] for k in fx if k.endswith("/reference_error"))
    assert rejected == ["pyramid_d0", "pyramid_d0", "pyramid_d1", "pyramid_d1"]


def test_step_fixture_rk():
    "oracle == reference RKUltraWrapperScheduler.step (Cash-Karp, SDE) bit for bit"
    fx = load_npz("steps_cfg5.npz")
    drv = OW.RKDriver(OK.pick_tableau(6), OS.scaled(), "eps", "data", 1.0)
    drv.set_timesteps(3)
    assert np.array_equal(drv.timesteps.numpy(), fx["timesteps"])
    noises = [torch.from_numpy(v) for v in fx["noises"][: int(fx["noise_used"])]]
    x = from_bits(fx["x0"], torch.bfloat16)
    for i, t in enumerate(drv.timesteps):
        out = drv.step(from_bits(fx["outs"][i], torch.bfloat16), t, x, noise_fn=lambda st: noises.pop(0))
        assert torch.equal(out, from_bits(fx["prev"][i], torch.bfloat16)), i
        x = out
    assert not noises


def test_noise_fixtures():
    "oracle noise stages == reference generators on the draws the reference consumed"
    fx = load_npz("noise.npz")
    tags = sorted({k.rsplit("/", 1)[0] for k in fx if k.endswith("/out")})
    assert len(tags) >= 20
    for tag in tags:
        kind, u = tag.split("/")
        unit = tuple(int(v) for v in u.split("x"))
        normals = [torch.from_numpy(fx[f"{tag}/normal{i}"]) for i in range(int(fx[f"{tag}/n_normals"]))]
        rp = ON.Replay(normals, fx[f"{tag}/uniforms"].tolist())
        if kind == "offset":
            got = ON.offset_noise(unit, rp.randn)
        elif kind == "offset_d02":
            got = ON.offset_noise(unit, rp.randn, (0, 2), 0.5)
        elif kind == "pyramid":
            got = ON.pyramid_noise(unit, rp.randn, rp.rand1)
        elif kind == "pyramid_depth1":
            got = ON.pyramid_noise(unit, rp.randn, rp.rand1, strength=0.6, depth=1)
        elif kind == "colored_energy":
            got = ON.colored_noise(unit, rp.randn, (0.3, 0.4), energy=2.5, color_start=1.5, color_end=-3, color_curve=0)
        else:
            step = [None, (0.0, 0.05), (0.45, 0.5), (0.95, 1.0)][int(kind[-1])]
            got = ON.colored_noise(unit, rp.randn, step)
        assert torch.equal(got, torch.from_numpy(fx[f"{tag}/out"])), tag
        assert not rp.normals
    for key in [k for k in fx if k.startswith("radial/")]:
        unit = tuple(int(v) for v in key.split("/")[1].split("x"))
        assert torch.equal(ON.radial_freq_grid(unit), torch.from_numpy(fx[key]))
    for a, b, e in fx["colored_exponents"]:
        assert ON.colored_exponent((a, b)) == e
    shapes = [tuple(r) for r in ON_levels((4, 128, 128), fx["pyramid_levels/4x128x128/uniforms"].tolist())]
    assert shapes == [tuple(r) for r in fx["pyramid_levels/4x128x128/shapes"][1:].tolist()]


def ON_levels(shape, uniforms):
    it = iter(uniforms)
    return [run for _, run, _ in ON.pyramid_levels(shape, lambda: next(it))]


def test_philox_known_answers():
    "Random123 kat_vectors for philox4x32-10"
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for c, k, e in kat:
        got = ON.philox4x32(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))
        assert tuple(int(v) for v in got) == e
    z = ON.philox_normal(42, 3 * 256, 1 << 18)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and abs((z**4).mean() - 3) < 0.1
    assert np.array_equal(ON.philox_normal(42, 5, 100, offset=37), ON.philox_normal(42, 5, 137)[37:])


def test_identities():
    "reference tests/self_sampling.py:107-171 (model transforms / convert / point round trips)"
    for pred in ("data", "eps", "flow", "v"):
        for sched in (OS.linear(), OS.scaled()):
            for eta in (-1.5, 0, 0.5, 1):
                p0 = sched.point(0.6)
                x = OP.to_x(pred, 0.8, 0.3, p0)
                assert abs(0.3 - OP.from_x(pred, 0.8, x, p0)) < 1e-12
                for t_next in (0.05, 0):
                    p1 = sched.point(t_next)
                    f = OP.forward(pred, 0.8, 0.3, p0, p1, 0.6, eta)
                    assert abs(f - OP.forward("data", 0.8, x, p0, p1, 0.6, eta)) < 1e-12
                    assert abs(0.3 - OP.backward(pred, 0.8, f, p0, p1, 0.6, eta)) < 1e-12
    for sigma in (1, 0.65, 0):
        for alpha in (1, 0.35, 0):
            for s in (-1.5, 0, 0.5, 1.5):
                for n in (-1.5, 0, 0.5, 1.5):
                    p = OC.Pt(sigma, sigma, alpha)
                    noisy = OC.pt_add_noise(p, s, n)
                    clean = OC.pt_remove_noise(p, noisy, n)
                    assert abs((s if alpha != 0 else noisy) - clean) < 1e-15


@pytest.mark.parametrize("tag", NATIVE16_TAGS)
def test_native_dtype_chain_equals_the_reference_bit_for_bit(tag):
    """Called directly on 16-bit tensors the reference's samplers compute in the tensor dtype, rounding after every torch op
    (structured.py:209-283).  The oracle restates that op order: its chain over the recorded inputs (tests/golden/native16.npz,
    outputs of the reference itself) reproduces every recorded `final` and `prediction` bit for bit, bf16 and fp16."""
    dt, steps, mname, sname, _, t = native16_case(load_npz("native16.npz"), tag)
    cfg, sched = NATIVE16_ORACLE[tag.split("/")[0]](OA), SCHEDULES[sname][0]()
    previous: list = []
    for i in range(steps):
        rec = OA.sample(cfg, t["x"][i], t["out"][i], (i / steps, (i + 1) / steps), MODELS[mname][0], sched, t["noise"][i] if OA.require_noise(cfg) else None, previous)
        assert rec.final.dtype == dt
        assert torch.equal(rec.final.view(torch.int16), t["final"][i].view(torch.int16)), (tag, i)
        assert torch.equal(torch.as_tensor(rec.prediction).view(torch.int16), t["prediction"][i].view(torch.int16)), (tag, i)
        previous.append(rec)
        keep = OA.require_previous(cfg)
        previous = previous[max(len(previous) - keep, 0) :] if keep else []


def test_colorize_with_up_to_six_axes_fixtures():
    "tests/golden/colorize_nd.npz (the reference's colorize_noise on 4-, 5- and 6-axis tensors): the oracle reproduces the recorded outputs"
    fx = load_npz("colorize_nd.npz")
    for tag in sorted({k.split("/")[0] for k in fx}):
        exponent, energy = fx[f"{tag}/args"].tolist()
        got = ON.colorize(torch.from_numpy(fx[f"{tag}/white"]), exponent, None if math.isnan(energy) else energy)
        assert torch.equal(got, torch.from_numpy(fx[f"{tag}/out"])), tag


# ---- the two reference-recorded sweeps (tests/golden/steps_sweep.npz, steps_sweep_native.npz) through the oracle --------------------------------
@pytest.mark.parametrize("fname", ["steps_sweep.npz", "steps_sweep_native.npz"])
def test_oracle_replays_the_reference_recorded_sweeps(fname):
    """every drawn configuration the oracle models (tests/oracle_sweep.py turns the constructor text into skr_oracle objects) -- teacher-forced, the reference's recorded outputs come back bit for bit: nested predictors, every schedule modifier, invert_prediction,
    compute_scale fp32 / fp64 / None on bf16 / fp16 / fp32 / fp64 tensors, the Runge-Kutta wrapper stage by stage"""
    import json

    import oracle_sweep as OSW
    from cases import from_bits

    blob = load_npz(fname)
    native = fname.endswith("native.npz")
    replayed = 0
    for m in json.loads(str(blob["meta"])):
        fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith(m["tag"] + "/")}
        dt = getattr(torch, m["dtype"])
        d = OSW.driver(m["text"], m["steps"])
        if d is None:
            continue
        np.testing.assert_allclose(d.timesteps.numpy(), fx["timesteps"], rtol=0, atol=1e-9)
        used = int(fx["noise_used"])
        noises = [from_bits(v, dt) if native else torch.from_numpy(v) for v in fx["noises"][:used]]
        x = from_bits(fx["x0"], dt)
        for i, t in enumerate(d.timesteps):
            out = from_bits(fx["outs"][i], dt)
            if isinstance(d, OSW.OW.StepDriver):
                prev, pred = d.step(out, t, x, noise=noises.pop(0) if used and OSW.OA.require_noise(d.cfg) else None)
                assert torch.equal(pred, from_bits(fx["pred"][i], dt)), (m["tag"], i, "pred_original_sample", m["text"])
            else:  # (what get_step_noise hands the stage is cast to compute_scale, or to the sample's dtype: diffusers.py:346)
                prev = d.step(out, t, x, noise_fn=lambda st: noises.pop(0).to(d.compute or dt))
            assert torch.equal(prev, from_bits(fx["prev"][i], dt)), (m["tag"], i, "prev_sample", m["text"])
            x = from_bits(fx["prev"][i], dt)
        replayed += 1
    assert replayed == (64 if not native else 32)


def test_oracle_replays_the_16_bit_functional_loops_and_transforms():
    "tests/golden/native_api16.npz through the oracle's loops and predictor functions: generic tensor operators on bf16 / fp16 tensors, the reference's bits"
    import json

    import oracle_sweep as OSW

    blob = load_npz("native_api16.npz")
    net = lambda xx, t, sg, al: xx * (0.3 - 0.1 * sg + 0.05 * al)  # noqa: E731
    checked = 0
    for m in json.loads(str(blob["meta"])):
        dt = torch.bfloat16 if m["dtype"] == "bf16" else torch.float16
        get = lambda key: torch.from_numpy(blob[f"{m['dtype']}/{key}"].copy()).view(dt)  # noqa: E731
        want = torch.from_numpy(blob[m["key"]].copy()).view(dt)
        if m["kind"] == "loop":
            pool = list(get("draws"))
            got = OSW.loop(m["sampler"], m["model"], m["schedule"], m["steps"], get("s").clone(), net, lambda *_: pool.pop(0))
        else:
            got = eval(m["text"], {**OSW.CALL_NAMES, "s": get("s"), "o": get("o"), "n": get("n")})
        assert got.dtype == dt and torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(want)), m
        checked += 1
    assert checked == 40
