"""Host-side logic of the product (skrample_amd): schedules, coefficient algebra, sampler step
plans -- exercised on Python scalars and symbolic operands, no GPU.  Written to read like the
reference's own tests (tests/self_sampling.py, tests/self_scheduling.py, tests/miscellaneous.py)."""

import itertools
import math
import random

import numpy as np
import pytest
import torch
from cases import MODELS, SAMPLERS, SCHEDULES, fake_model
from conftest import eq_nan

import skrample_amd.diffusers as PD
import skrample_amd.scheduling as PS
from skr_oracle import rk as OK
from skr_oracle import samplers as OA
from skr_oracle import schedules as OS
from skrample_amd.common import DeltaPoint, MergeStrategy, Point, Step, bashforth, sigmoid, softmax, spowf
from skrample_amd.sampling import functional as PF
from skrample_amd.sampling import interface as PI
from skrample_amd.sampling import lazy
from skrample_amd.sampling import models as PM
from skrample_amd.sampling import structured as PT
from skrample_amd.sampling import tableaux as PTab


def product_trajectory(sampler, schedule, model, steps=7, seed=42):
    out = []
    random.seed(seed)
    (PI.StructuredFunctionalAdapter(sampler) if isinstance(sampler, PT.StructuredSampler) else sampler).generate_model(
        fake_model, model, PS.Hyper(schedule), lambda _: random.random(), steps, callback=lambda x, i, d: out.append(x)
    )
    return out


def oracle_trajectory(cfg, sched, pred, steps=7, seed=42):
    out = []
    random.seed(seed)
    hs = OS.hyper(sched)
    loop = lambda x: OA.adapter_loop(cfg, x, fake_model, pred, hs, steps, rng=lambda _: random.random(), callback=lambda x, i, d: out.append(x))  # noqa: E731
    OA.generate(loop, None, lambda _: random.random(), hs, steps)
    return out


def test_reference_sampler_goldens(kats):
    "the 24 committed reference trajectories (reference tests/self_sampling.py:57-104, tolerance 1e-3 %)"
    mk = {"RKUltra": lambda: PF.RKUltra(providers={2: PTab.RKE2.Heun}), "DynasauRK": PF.DynasauRK, "Adams": PT.Adams, "SPC": PT.SPC}
    for key, ref in kats["sampler_trajectories"].items():
        s, sch, m = key.split("/")
        got = product_trajectory(mk[s](), getattr(PS, sch)(), getattr(PM, m)())
        np.testing.assert_allclose(got, ref, rtol=1e-9, err_msg=key)


def test_reference_schedule_goldens(kats):
    env = {k: getattr(PS, k) for k in ("Linear", "Scaled", "Karras", "Beta", "Exponential", "Probit", "Hyper", "Sinner")}
    for label, ref in kats["schedule_points"].items():
        got = eval(label, {"__builtins__": {}}, env).points_np(np.linspace(1, 0, 7))  # noqa: S307
        np.testing.assert_allclose(got, np.asarray(ref), rtol=1e-5, err_msg=label)


@pytest.mark.parametrize("name", SCHEDULES)
def test_schedules_equal_oracle(name):
    o, p = SCHEDULES[name][0](), SCHEDULES[name][1]()
    t = [0, 1, *np.random.default_rng(0).random(30)]
    assert eq_nan(o.points_np(t), p.points_np(t))
    assert eq_nan(o.schedule_np(13), p.schedule_np(13))
    assert tuple(o.ipoint(0.3)) == tuple(p.ipoint(0.3))
    hash(p)  # schedules key LRU caches
    if "zsnr" not in name and "neg" not in name:
        assert p.point(0) == (0, 0, 1)  # reference test_zero_point (Linear / Scaled bases)


def test_wrapper_tables(tables):
    for key, ref in tables["wrapper"].items():
        name, n = key.split("/")
        w = PD.SkrampleWrapperScheduler(PT.Euler(), SCHEDULES[name][1]())
        w.set_timesteps(int(n))
        assert eq_nan(w.timesteps.tolist(), ref["timesteps"]), key
        assert eq_nan(w.sigmas.tolist(), ref["sigmas"]), key
        assert eq_nan(w.schedule_np.tolist(), ref["schedule_np"]), key


def test_gamma_delta_zeta(tables):
    for row in tables["gdz"]:
        sched = SCHEDULES[row["schedule"]][1]()
        dp = DeltaPoint(*sched.ipoints(row["step"]))
        m = MODELS[row["model"]][1]
        try:
            got = [m.gamma(dp, row["eta"]), m.delta(dp, row["eta"]), m.zeta(dp, row["eta"])]
        except ZeroDivisionError:
            got = None
        ref = row["gdz"]
        if ref is None or got is None:
            assert ref is None and got is None, row
        elif isinstance(ref[0], str):
            assert [repr(v) for v in got] == ref, row
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-15, atol=0, err_msg=str(row))


def test_effective_order(tables):
    kind = {"DPM": PT.DPM, "Adams": PT.Adams, "UniP": PT.UniP, "UniPC": PT.UniPC}
    for row in tables["effective_order"]:
        s = kind[row["sampler"]](order=row["order"])
        got = [s.effective_order(Step.from_int(i, row["steps"]), [None] * row["n_previous"]) for i in range(row["steps"])]
        assert got == row["eo"] and s.require_previous == row["require_previous"], row


def test_tableaux(tables):
    groups = (PTab.RK1, PTab.RK2, PTab.RK3, PTab.RK4, PTab.RKZ, PTab.RKE2, PTab.RKE3, PTab.RKE5, PTab.SSP, PTab.WSO, PTab.Shanks1965)
    for grp in groups:
        for member in grp:
            ref = tables["tableaux"][f"{grp.__name__}.{member.name}"]
            tab = member.tableau()
            assert PTab.validate_tableau(tab) is None, member  # reference test_tableau_providers
            flat_ref = [*ref["c"], *(v for r in ref["a"] for v in r), *ref["b"]]
            np.testing.assert_allclose(PTab.serialize(tab), flat_ref, rtol=0, atol=1e-15, err_msg=str(member))
    for k, v in tables["default_providers"].items():
        if True:
            grp, name = v.split(".")
            assert PF.DEFAULT_PROVIDERS[int(k)] is getattr(getattr(PTab, grp), name)
            assert int(k) == len(PF.DEFAULT_PROVIDERS[int(k)].tableau()[0])  # reference test_tableau_preset_stages
    for k, v in tables["stable_providers"].items():
        grp, name = v.split(".")
        assert PF.STABLE_PROVIDERS[int(k)] is getattr(getattr(PTab, grp), name)
    assert len(PF.RKUltra(order=99).tableau().stages) == 15  # Stepanov10
    assert [m.name for m in PTab.GRAVEYARD] == [m.name for m in (*PTab.WSO, *PTab.Shanks1965)] and len(PTab.GRAVEYARD) == 14
    assert not set(PTab.GRAVEYARD) & set(PTab.BUILTIN_TABLEAUX)  # kept, not recommended (reference tableaux/__init__.py:39-43)


def test_rk_points(tables):
    for key, ref in tables["rk_points"].items():
        kind, name, order, steps = key.split("/")
        if kind == "rku":
            w = PD.RKUltraWrapperScheduler(SCHEDULES[name][1](), sampler_order=int(order))
        else:
            w = PD.DynasauRKWrapperScheduler(SCHEDULES[name][1](), sampler_order=int(order), model=PM.FlowModel() if "linear" in name else PM.NoiseModel())
        w.set_timesteps(int(steps))
        np.testing.assert_allclose([list(p) for p in w.all_points], ref["all"], rtol=0, atol=1e-12, err_msg=key)
        np.testing.assert_allclose(w.timesteps.tolist(), ref["timesteps"], rtol=0, atol=1e-9)
        assert w.order == ref["order"] and len(w.timesteps) <= int(steps) * w.order


@pytest.mark.parametrize("sampler", SAMPLERS)
def test_samplers_equal_oracle_on_scalars(sampler):
    "every sampler x schedule x model: product step algebra == oracle (== reference), 9-step trajectories"
    mk_o, mk_p = SAMPLERS[sampler]
    for sname in ("linear", "scaled", "zsnr", "karras_scaled"):
        for mname in ("data", "flow", "v", "eps", "scalex"):
            if mname == "eps" and sname in ("linear", "zsnr"):
                continue  # alpha = 0 at t = 1: division by zero in the reference too
            a = oracle_trajectory(mk_o(), SCHEDULES[sname][0](), MODELS[mname][0], steps=9)
            b = product_trajectory(mk_p(), SCHEDULES[sname][1](), MODELS[mname][1], steps=9)
            np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-12, err_msg=f"{sampler}/{sname}/{mname}")
    assert OA.require_noise(mk_o()) == mk_p().require_noise
    assert OA.require_previous(mk_o()) == mk_p().require_previous


def test_spc_power_scalar():
    a = oracle_trajectory(OA.make("spc", power=2), OS.scaled(), "v", steps=9)
    b = product_trajectory(PT.SPC(power=2), PS.Scaled(), PM.VelocityModel(), steps=9)
    np.testing.assert_allclose(b, a, rtol=1e-9)


@pytest.mark.parametrize("order", [1, 2, 3, 4, 5, 6])
def test_rkultra_equals_oracle(order):
    for (so, sp), (po, pp), eta in itertools.product([SCHEDULES["linear"], SCHEDULES["scaled"]], [MODELS["data"], MODELS["flow"], MODELS["v"]], (0, 0.7)):
        random.seed(1)
        a = OK.rk_loop(lambda st: OK.pick_tableau(order), 0.7, fake_model, po, OS.hyper(so()), 5, rng=lambda _: random.random(), eta=eta)
        random.seed(1)
        b = PF.RKUltra(order=order, stochasticity=eta).sample_model(0.7, fake_model, pp, PS.Hyper(sp()), 5, rng=lambda _: random.random())
        assert abs(a - b) < 1e-10 * max(1, abs(a))


# ---- identities the reference tests pin (tests/self_sampling.py:107-171, 332-414) ----------------------
@pytest.mark.parametrize(("model", "schedule", "eta"), itertools.product([PM.DataModel, PM.NoiseModel, PM.FlowModel, PM.VelocityModel], [PS.Linear, PS.Scaled], [-1.5, 0, 0.5, 1]))
def test_model_transforms(model, schedule, eta):
    m = model()
    sample, output, noise = 0.8, 0.3, 0.6
    p0 = schedule().point(0.6)
    x = m.to_x(sample, output, p0)
    assert abs(output - m.from_x(sample, x, p0)) < 1e-12
    for t_next in (0.05, 0):
        delta = DeltaPoint(p0, schedule().point(t_next))
        f = m.forward(sample, output, delta, noise, eta)
        assert abs(f - PM.DataModel().forward(sample, x, delta, noise, eta)) < 1e-12
        assert abs(output - m.backward(sample, f, delta, noise, eta)) < 1e-12


@pytest.mark.parametrize(
    ("m_from", "m_to", "schedule", "sigma_to"),
    itertools.product([PM.DataModel, PM.NoiseModel, PM.FlowModel, PM.VelocityModel], [PM.DataModel, PM.NoiseModel, PM.FlowModel, PM.VelocityModel, PM.ScaleX], [PS.Linear, PS.Scaled], (0.05, 0.0)),
)
def test_model_convert(m_from, m_to, schedule, sigma_to):
    conv = PM.ModelConvert(m_from(), m_to())
    delta = DeltaPoint(schedule().point(0.2), schedule().point(sigma_to))
    model = lambda x, t, s, a: 0.3  # noqa: E731
    a = conv.transform_from.forward(0.8, model(0.8, *delta.point_from), delta)
    b = conv.transform_to.forward(0.8, conv.wrap_model_call(model)(0.8, *delta.point_from), delta)
    assert abs(a - b) < 1e-12


@pytest.mark.parametrize(("sigma", "alpha", "sample", "noise"), itertools.product([1, 0.65, 0], [1, 0.35, 0], [-1.5, 0, 0.5, 1.5], [-1.5, 0, 0.5, 1.5]))
def test_point(sigma, alpha, sample, noise):
    p = Point(sigma, sigma, alpha)
    noisy = p.add_noise(sample, noise)
    clean = p.remove_noise(noisy, noise)
    assert abs((sample if alpha != 0 else noisy) - clean) < 1e-15


@pytest.mark.parametrize(("model", "schedule", "noise"), itertools.product([PM.DataModel, PM.NoiseModel, PM.VelocityModel, PM.FlowModel], [PS.Sinner(PS.Linear()), PS.Scaled()], [False, True]))
def test_maruyama(model, schedule, noise):
    "DPM(order=1, eta) == Euler(eta) step for step (reference test_maruyama)"
    if model is PM.NoiseModel and isinstance(schedule.space, PS.FlowMatching):
        return
    dpm = PI.StructuredFunctionalAdapter(PT.DPM(order=1, stochasticity=noise))
    eul = PI.StructuredFunctionalAdapter(PT.Euler(stochasticity=int(noise)))
    f = lambda x, t, s, a: x + math.sin(x) * s  # noqa: E731
    random.seed(0)
    a = dpm.sample_model(1.7, f, model(), schedule, 23, rng=lambda _: random.random())
    random.seed(0)
    b = eul.sample_model(1.7, f, model(), schedule, 23, rng=lambda _: random.random())
    assert abs(a - b) < 1e-12


@pytest.mark.parametrize(("sampler", "steps"), itertools.product([PT.DPM(o, n) for o in range(1, 4) for n in (False, True)], [1, 3, 4, 9, 512]))
def test_functional_adapter(sampler, steps):
    "adapter loop == manual sampler.sample loop, exactly (reference test_functional_adapter)"
    f = lambda x, t, s, a: x + math.sin(x) * s  # noqa: E731
    schedule, transform = PS.Scaled(), PM.FlowModel()
    noise = [random.random() for _ in range(steps)]
    rng = iter(noise)
    via_adapter = PI.StructuredFunctionalAdapter(sampler).sample_model(1.5, f, transform, schedule, steps, rng=lambda _: next(rng))
    rng = iter(noise)
    x, prev = 1.5, []
    for n, (t, s, a) in enumerate(schedule.schedule(steps)):
        rec = sampler.sample(x, f(x, t, s, a), Step.from_int(n, steps), transform, schedule, next(rng), prev)
        prev.append(rec)
        x = rec.final
    assert x == via_adapter


def test_require_previous_and_noise():
    "history may be trimmed to require_previous; noise matters iff require_noise (reference :227-329)"
    previous = tuple(PT.SKSamples(n / 2, n * 2, Step.from_int(n, 100), 1 / (n + 1), n * 1.5) for n in range(31))
    samplers = [mk() for _, mk in SAMPLERS.values()]
    for s in samplers:
        a = s.sample(1.5, 0.5, Step.from_int(31, 100), PM.DataModel(), PS.Linear(), None, previous)
        b = s.sample(1.5, 0.5, Step.from_int(31, 100), PM.DataModel(), PS.Linear(), None, previous[len(previous) - s.require_previous :])
        assert a.final == b.final, s
        c = s.sample(1.5, 0.5, Step.from_int(31, 100), PM.DataModel(), PS.Linear(), -0.5, previous)
        d = s.sample(1.5, 0.5, Step.from_int(31, 100), PM.DataModel(), PS.Linear(), None, [PT.SKSamples(p.sample, p.prediction, p.step, None, p.final) for p in previous])
        assert (c.final == d.final) ^ s.require_noise, s


def test_misc_common():
    "reference tests/miscellaneous.py"
    for n, coeffs in enumerate(((1,), (3 / 2, -1 / 2), (23 / 12, -4 / 3, 5 / 12), (55 / 24, -59 / 24, 37 / 24, -3 / 8))):
        assert np.allclose(coeffs, bashforth(n + 1), atol=1e-12, rtol=1e-12)
    items = [spowf(v, 2) for v in np.linspace(-2, 2, 9).tolist()]
    assert np.allclose([sigmoid(v) for v in items], torch.sigmoid(torch.tensor(items, dtype=torch.float64)).tolist(), atol=1e-12)
    assert np.allclose(softmax(tuple(items)), torch.softmax(torch.tensor(items, dtype=torch.float64), 0).tolist(), atol=1e-12)
    a, b = list(range(0, 11)), list(range(0, 15, 2))
    assert MergeStrategy.UniqueBefore.merge(a, b) == b + list(range(1, 10, 2))
    assert MergeStrategy.UniqueAfter.merge(a, b) == a + list(range(12, 15, 2))
    assert MergeStrategy.Before.merge(a, b) == b + a and MergeStrategy.Ours.merge(a, b) == a
    for n in range(32):
        st = Step.from_int(n, 31)
        assert abs(st.amount() - 31) < 1e-8 and abs(st.position() - n) < 1e-8 and Step(*reversed(st)).normal() == st
        assert abs(st.offset(-4).position() - (n - 4)) < 1e-8
        assert st.offset(15.5).clamp().position() + 1 <= 31 + 1e-8 and st.offset(-15.5).clamp().position() >= 0
    for cls, data in PD.DIFFUSERS_CLASS_MAP.values():
        cls(**data)


def test_config_parsing():
    "reference tests/diffusers_map.py (the cases that need no diffusers install)"
    flow = {"base_shift": 0.5, "flow_shift": 3.0, "num_train_timesteps": 1000, "prediction_type": "flow_prediction", "shift": 3.0, "use_dynamic_shifting": True}
    w = PD.SkrampleWrapperScheduler.from_diffusers_config(flow)
    assert w.sampler == PT.DPM() and w.model == PM.FlowModel() and w.schedule == PS.FlowShift(PS.Linear(), shift=3.0)
    scaled = {"beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085, "num_train_timesteps": 1000, "prediction_type": "epsilon", "use_karras_sigmas": True, "_class_name": "EulerAncestralDiscreteScheduler"}
    w = PD.SkrampleWrapperScheduler.from_diffusers_config(scaled)
    assert w.sampler == PT.Euler(stochasticity=True) and w.model == PM.NoiseModel() and w.schedule == PS.Karras(PS.Scaled(beta_scale=2))
    w = PD.SkrampleWrapperScheduler.from_diffusers_config({"_class_name": "UniPCMultistepScheduler", "solver_order": 3, "prediction_type": "v_prediction", "rescale_betas_zero_snr": True})
    assert w.sampler == PT.UniPC(order=3) and w.model == PM.VelocityModel() and isinstance(w.schedule, PS.ZSNR)
    w = PD.SkrampleWrapperScheduler.from_diffusers_config({"_class_name": "MiniMaxH3Scheduler", "shift": 12})
    assert w.invert_prediction and w.schedule.lowest.base_timesteps == -1
    assert w.config["shift"] == 12 and w.config.prediction_type == "flow"
    a = PD.SkrampleWrapperScheduler(PT.DPM(), PS.Hyper(PS.FlowShift(PS.Hyper(PS.Linear()))))
    a.set_timesteps(123, mu=1.2345)  # reference test_mu_set
    assert a.schedule == PS.Hyper(PS.FlowShift(PS.Hyper(PS.Linear()), shift=math.exp(1.2345)))
    k = PD.SkrampleWrapperScheduler(PT.DPM(), PS.Karras(PS.Scaled()))
    k.set_timesteps(31)
    assert k.schedule == PS.Karras(PS.Scaled(), steps=31) and len(k.timesteps) == 31 and len(k.sigmas) == 32
    r = PD.RKUltraWrapperScheduler.from_diffusers_config(flow, sampler_order=6)
    r.set_timesteps(4)
    assert r.order == 6 and isinstance(r.model, PM.FlowModel)
    # which sub-schedule wins when a diffusers config raises several `use_*_sigmas` flags (reference tests/diffusers_map.py:162-230, the table
    # itself; their configs come from a hub download, these are the same keys spelled out)
    vp = {"_class_name": "DPMSolverMultistepScheduler", "beta_schedule": "scaled_linear", "beta_start": 0.00085, "beta_end": 0.012, "num_train_timesteps": 1000, "prediction_type": "epsilon"}
    fm = {"_class_name": "DPMSolverMultistepScheduler", "num_train_timesteps": 1000, "prediction_type": "flow_prediction", "use_flow_sigmas": True, "flow_shift": 3}
    table_vp = {(1, 1, 1): PS.Karras, (0, 1, 1): PS.Exponential, (1, 0, 1): PS.Karras, (1, 1, 0): PS.Karras, (1, 0, 0): PS.Karras, (0, 1, 0): PS.Exponential, (0, 0, 1): PS.Beta, (0, 0, 0): None}
    table_fm = {(1, 1, 1): PS.FlowShift, (0, 1, 1): PS.FlowShift, (1, 0, 1): PS.FlowShift, (1, 1, 0): PS.FlowShift, (1, 0, 0): PS.FlowShift, (0, 1, 0): PS.FlowShift, (0, 0, 1): PS.Beta, (0, 0, 0): PS.FlowShift}
    for (karras, exp, beta), sub in table_vp.items():
        flags = {"use_karras_sigmas": bool(karras), "use_exponential_sigmas": bool(exp), "use_beta_sigmas": bool(beta), "use_flow_sigmas": False, "flow_shift": 3}
        w = PD.SkrampleWrapperScheduler.from_diffusers_config(vp | flags)
        assert w.sampler == PT.DPM() and w.schedule == (PS.Scaled() if sub is None else sub(PS.Scaled())), (karras, exp, beta, w.schedule)
    for (karras, exp, beta), sub in table_fm.items():
        flags = {"use_karras_sigmas": bool(karras), "use_exponential_sigmas": bool(exp), "use_beta_sigmas": bool(beta)}
        w = PD.SkrampleWrapperScheduler.from_diffusers_config(fm | flags)
        assert w.schedule == sub(PS.Linear()) and w.model == PM.FlowModel(), (karras, exp, beta, w.schedule)


def test_lazy_algebra_is_symbolic():
    "the sampler algebra never touches tensor data: forms over opaque leaves just accumulate coefficients"

    class Opaque(lazy.PhiloxNoise):  # any leaf object works; reuse the symbolic-noise leaf type
        def __init__(self, tag):
            super().__init__(None, 0, (2, 8), torch.device("cpu"))
            self.tag = tag

    x, o, xp, op_, n = (lazy.Lin.leaf(Opaque(t)) for t in "x o xp op n".split())
    sched, model = PS.Karras(PS.Scaled()), PM.NoiseModel()
    prev = [PT.SKSamples(xp, op_, Step.from_int(4, 20), None, None)]
    form = PT.DPM(order=2, stochasticity=1)._form(PT.SampleInput(x, o, Step.from_int(5, 20), n), model, sched, prev)
    coef = {leaf.tag: c for leaf, c in form.terms.values()}
    assert set(coef) == {"x", "o", "xp", "op", "n"}
    # same numbers from the oracle on unit "tensors": coefficient of leaf k = f(e_k)
    from skr_oracle import schedules as OS

    osched = OS.karras(OS.scaled())
    cfg = OA.make("dpm", 2, eta=1)
    for k, tag in enumerate(("x", "o", "xp", "op", "n")):
        e = [1.0 if i == k else 0.0 for i in range(5)]
        rec = OA.sample_packed(cfg, OA.Rec(e[0], e[1], (5 / 20, 6 / 20), e[4]), "eps", osched, [OA.Rec(e[2], e[3], (4 / 20, 5 / 20))])
        assert abs(rec.final - coef[tag]) < 1e-12 * max(1, abs(coef[tag])), tag
    with pytest.raises(lazy.SkrampleHipError):
        x * o
    # host-resident operands are leaves too (evaluated by the host executor): numpy forms remember where they came from
    assert lazy.lift(np.ones(3)).device == lazy.NUMPY and lazy.lift(torch.ones(3)).device.type == "cpu"
    arr = np.ones(3)
    assert list(lazy.lift(arr).terms) == list(lazy.lift(arr).terms)  # one leaf per array, however often it is lifted
    with pytest.raises(lazy.SkrampleHipError):
        lazy.lift(np.ones(3, dtype=np.int64))
    with pytest.raises(lazy.SkrampleHipError):
        lazy.lift("not a tensor")


# ---- schedule behaviour the reference pins in tests/self_scheduling.py:49-151 -----------------------------------
ALL_BASES = [PS.Linear, PS.Scaled, lambda **k: PS.Scaled(beta_scale=1, **k)]
ALL_MODIFIERS = [None, PS.NoSub, PS.NoMod, PS.Beta, PS.FlowShift, PS.Karras, PS.Exponential, PS.Probit, PS.Hyper, PS.Sinner]


@pytest.mark.parametrize(("base", "modifier"), itertools.product(ALL_BASES, ALL_MODIFIERS))
def test_schedule_properties(base, modifier):
    sched = modifier(base()) if modifier else base()
    t100 = [0, 1, *np.random.default_rng(1).random(98)]
    batch = sched.points_np(t100)
    single = np.array([sched.point(t) for t in t100], dtype=np.float64)
    assert np.array_equal(batch, single)  # continuously variable: batch == point by point, exactly
    assert sched.point(0) == (0, 0, 1)  # zero point
    if modifier:
        np.testing.assert_allclose(sched.point(1), base().point(1), rtol=0, atol=1e-15)  # modifiers keep the t=1 point
    for steps in (1, 2, 3, 999, 1000, 1001):  # timestep inversion with negative base_timesteps
        fwd = modifier(base(base_timesteps=steps)) if modifier else base(base_timesteps=steps)
        bwd = modifier(base(base_timesteps=-steps)) if modifier else base(base_timesteps=-steps)
        pts = np.linspace(0, 1, steps)
        a, b = fwd.points_np(pts).copy(), bwd.points_np(pts).copy()
        b[:, 0] = steps - b[:, 0]
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-12)


@pytest.mark.parametrize("base", ALL_BASES)
def test_sigmas_to_points(base):
    sched = base()
    pts = sched.points_np(np.linspace(1, 0, 33))
    inv = sched._sigmas_to_points(pts[:, 1], pts[:, 2])
    for _ in range(99):
        inv = sched._sigmas_to_points(inv[:, 1], inv[:, 2])
    dev = np.abs(pts - inv)
    assert (dev <= 0.001 * np.abs(pts) + 1e-12).all()  # within 0.1 %, as the reference asks


@pytest.mark.parametrize(("modifier", "timesteps", "steps"), itertools.product([None, PS.Karras, PS.FlowShift, PS.Hyper], [1, 999, 1000, 1001, -1001, -1], [1, 999, 1002]))
def test_terminal_timesteps(modifier, timesteps, steps):
    "every wrapper exposes steps * order timesteps (reference test_terminal_timesteps)"
    sched = modifier(PS.Linear(base_timesteps=abs(timesteps))) if modifier else PS.Linear(base_timesteps=abs(timesteps))
    for w in (
        PD.SkrampleWrapperScheduler(PT.Euler(), sched, model=PM.FlowModel()),
        PD.RKUltraWrapperScheduler(sched, sampler_order=1, model=PM.FlowModel()),
        PD.DynasauRKWrapperScheduler(sched, sampler_order=1, model=PM.FlowModel()),
    ):
        w.set_timesteps(steps)
        assert len(w.timesteps) == steps * w.order


@pytest.mark.parametrize("order", [2, 4, 6, 99])
def test_rkmoire_equals_oracle_on_scalars(order):
    "adaptive RK (reference functional.py:352-472): same accepted steps and result as the oracle"
    f = lambda x, t, s, a: x + math.sin(x) * s  # noqa: E731
    for (so, sp), (po, pp), steps in itertools.product([SCHEDULES["linear"], SCHEDULES["scaled"]], [MODELS["data"], MODELS["flow"], MODELS["v"]], (10, 37)):
        for kw in ({}, dict(threshold=1e-2, adaption=0.5, discard=1.5), dict(rescale_max=True, maximum=0.5)):
            seen_o, seen_p = [], []
            a = OK.rkmoire_loop(1.3, f, po, so(), steps, order=order, callback=lambda x, i, d: seen_o.append(i), **kw)
            b = PF.RKMoire(order=order, **kw).sample_model(1.3, f, pp, sp(), steps, callback=lambda x, i, d: seen_p.append(i))
            assert seen_o == seen_p and abs(a - b) < 1e-9 * max(1, abs(a)), (order, steps, kw)


def test_brownian_tree_weights_follow_the_brownian_law():
    """W(t) is a linear function of independent node normals, so Var/Cov are dot products of the weight
    vectors: Var W(t) = t, Cov(W(s), W(t)) = min(s, t), disjoint increments uncorrelated, adjacent increments add."""
    import random

    from skrample_amd.pytorch.noise import brownian_depth, brownian_increment, brownian_path

    depth = brownian_depth(10_000)
    assert depth == 19
    dot = lambda a, b: sum(v * b.get(k, 0.0) for k, v in a.items())
    rng = random.Random(7)
    leaf = 2.0**-depth
    for _ in range(50):
        s, t = sorted((rng.random(), rng.random()))
        ws, wt = brownian_path(s, depth), brownian_path(t, depth)
        assert abs(dot(ws, ws) - s) <= leaf and abs(dot(wt, wt) - t) <= leaf and abs(dot(ws, wt) - s) <= leaf
        assert len(wt) <= depth + 1
    assert brownian_path(0.0, depth) == {0: 0.0} and brownian_path(1.0, depth) == {0: 1.0}
    assert brownian_path(0.5, depth) == {0: 0.5, 1: 0.5}  # dyadic times stop early
    for n in (7, 20, 50, 1000):
        k = n // 3
        inc = [dict(zip(*brownian_increment(j / n, (j + 1) / n, depth))) for j in (k, k + 1, k + 3)]
        for w in inc:
            assert abs(dot(w, w) - 1) <= leaf * n and len(w) <= 2 * depth + 1 <= 64
        assert abs(dot(inc[0], inc[1])) <= leaf * n and abs(dot(inc[0], inc[2])) <= 1e-12 + leaf * n
        # additivity: sqrt(dt) * (inc_k + inc_{k+1}) is the increment over the union
        both = dict(zip(*brownian_increment(k / n, (k + 2) / n, depth)))
        for node in {*both, *inc[0], *inc[1]}:
            lhs = (inc[0].get(node, 0.0) + inc[1].get(node, 0.0)) * math.sqrt(1 / n)
            assert abs(lhs - both.get(node, 0.0) * math.sqrt(2 / n)) < 1e-12


def test_brownian_partition_path_follows_the_brownian_law():
    """round 4: a generator first asked one cell of an N-cell partition builds its path over that partition (index bisection for the grid
    points, a dyadic bridge inside a cell).  Same law, exactly on the grid -- Var W(j/N) = j/N, Cov = min -- and to the leaf width off it;
    grid queries need ceil(log2 N) + 1 normals per endpoint instead of depth + 1 = 20."""
    import random

    from skrample_amd.common import Step
    from skrample_amd.pytorch.noise import brownian_depth, brownian_endpoints, brownian_grid_of, brownian_grid_path, brownian_grid_point

    depth = brownian_depth(10_000)
    dot = lambda a, b: sum(v * b.get(k, 0.0) for k, v in a.items())
    for n in (1, 2, 7, 20, 30, 50, 1000):
        pts = {j: brownian_grid_point(j, n) for j in range(0, n + 1, max(1, n // 25))}
        for j, w in pts.items():
            assert len(w) <= (n - 1).bit_length() + 1
            for i, v in pts.items():
                assert abs(dot(w, v) - min(i, j) / n) < 1e-15
        for k in range(n):
            assert brownian_grid_of(*Step.from_int(k, n)) == n  # what a sampling run asks for
    assert brownian_grid_of(0.35, 0.45) is None and brownian_grid_of(0.1234, 0.2) is None and brownian_grid_of(0.4, 0.4) is None
    rng = random.Random(3)
    leaf = 4 * 2.0**-depth
    for n in (20, 30):
        for _ in range(100):
            s, t = sorted((rng.random(), rng.random()))
            ws, wt = brownian_grid_path(s, n, depth), brownian_grid_path(t, n, depth)
            assert abs(dot(ws, ws) - s) <= leaf and abs(dot(wt, wt) - t) <= leaf and abs(dot(ws, wt) - s) <= leaf and len(wt) <= 64
            j = int(t * n)  # consistency with the grid: Cov(W(t), W(j/n)) = j/n
            assert abs(dot(wt, brownian_grid_point(j, n)) - j / n) <= leaf
    # a step of a 20-step run: 5 normals for both endpoints together, 36 on the dyadic tree; additivity over adjacent cells
    nodes, w_to, w_from = brownian_endpoints(0.35, 0.4, depth, 20)
    assert len(nodes) <= 6 and len(brownian_endpoints(0.35, 0.4, depth, None)[0]) > 30
    a = {k: x - y for k, x, y in zip(nodes, w_to, w_from)}
    nodes, w_to, w_from = brownian_endpoints(0.4, 0.45, depth, 20)
    b = {k: x - y for k, x, y in zip(nodes, w_to, w_from)}
    nodes, w_to, w_from = brownian_endpoints(0.35, 0.45, depth, 20)
    ab = {k: x - y for k, x, y in zip(nodes, w_to, w_from)}
    for k in {*a, *b, *ab}:
        assert abs(a.get(k, 0.0) + b.get(k, 0.0) - ab.get(k, 0.0)) < 1e-15
    assert abs(dot(a, a) - 0.05) < 1e-15 and abs(dot(a, b)) < 1e-15


def test_brownian_oracle_is_an_independent_restatement():
    "the oracle bisects on arrays, the product walks weights: same function of the same Philox normals"
    from skr_oracle import noise as ON
    from skrample_amd.pytorch.noise import BROWNIAN_STREAMS, brownian_depth, brownian_increment

    depth, n = brownian_depth(10_000), 64
    for t0, t1 in ((0.35, 0.4), (0.0, 0.05), (0.95, 1.0), (0.5, 0.75)):
        nodes, weights = brownian_increment(t0, t1, depth)
        mine = sum(w * ON.philox_normal(5, BROWNIAN_STREAMS | h, n).astype(np.float64) for h, w in zip(nodes, weights))
        ref = ON.brownian_noise(5, (n,), (t0, t1)).numpy()
        assert np.abs(mine - ref).max() < 1e-9
    from skrample_amd.pytorch.noise import brownian_endpoints

    for t0, t1 in ((0.35, 0.4), (0.0, 0.05), (0.95, 1.0), (0.5, 0.75), (0.35, 0.45), (0.123, 0.777), (0.36, 0.39)):  # the partition path (grid = 20)
        nodes, w_to, w_from = brownian_endpoints(t0, t1, depth, 20)
        mine = sum((a - b) * ON.philox_normal(5, BROWNIAN_STREAMS | h, n).astype(np.float64) for h, a, b in zip(nodes, w_to, w_from)) / math.sqrt(t1 - t0)
        assert np.abs(mine - ON.brownian_noise(5, (n,), (t0, t1), grid=20).numpy()).max() < 1e-9
    # reference call-site arithmetic: direction-normalised and clamped steps (noise.py:241)
    assert torch.equal(ON.brownian_noise(5, (n,), (0.4, 0.35)), ON.brownian_noise(5, (n,), (0.35, 0.4)))
    assert torch.equal(ON.brownian_noise(5, (n,), (1.0, 1.05)), ON.brownian_noise(5, (n,), (0.95, 1.0)))


def test_tableau_constructors():
    "reference tests/self_sampling.py:675-800: the parametric constructors reproduce the published tableaux"
    T, S, B = PTab.common.Tableau, PTab.common.Stage, PTab.common.ButcherCoeffs

    def distance(a, b) -> float:
        return max(abs(x - y) for x, y in zip(PTab.serialize(a), PTab.serialize(b)))

    assert distance(T((S(0.0, ()), S(2 / 3, (2 / 3,))), (1 / 4, 3 / 4)), PTab.providers.rk2_tableau(2 / 3)) < 1e-20  # Ralston
    wray = T((S(0.0, ()), S(8 / 15, (8 / 15,)), S(2 / 3, (1 / 4, 5 / 12))), (1 / 4, 0.0, 3 / 4))
    assert distance(wray, PTab.providers.rk3_tableau(8 / 15, 2 / 3)) < 1e-15
    eighth = T((S(0, ()), S(1 / 3, (1 / 3,)), S(2 / 3, (-1 / 3, 1)), S(1, (1, -1, 1))), (1 / 8, 3 / 8, 3 / 8, 1 / 8))
    assert distance(eighth, PTab.providers.rk4_tableau(1 / 3, 2 / 3)) < 1e-12
    ees25 = T((S(0, ()), S(1 / 3, (1 / 3,)), S(5 / 6, (-5 / 48, 15 / 16))), (1 / 10, 1 / 2, 2 / 5))  # arXiv 2507.21006 (8.4)
    assert distance(ees25, PTab.providers.ees25_tableau(1 / 10)) < 1e-15
    v2 = math.sqrt(2)
    ees27 = T(
        (
            S(0, ()),
            S(1 / 3 * (2 - v2), (1 / 3 * (2 - v2),)),
            S(1 / 6 * (2 + v2), (1 / 24 * (-4 + v2), 1 / 8 * (4 + v2))),
            S(1 / 6 * (4 + v2), (1 / 168 * (-176 + 145 * v2), 3 / 56 * (8 - 5 * v2), 3 / 7 * (3 - v2))),
        ),
        (1 / 14 * (5 - 3 * v2), 1 / 14 * (3 + v2), 3 / 14 * (-1 + 2 * v2), 1 / 14 * (9 - 4 * v2)),
    )
    assert distance(ees27, PTab.providers.ees27_tableau(1 / 14 * (5 - 3 * v2))) < 1e-15
    ssp45 = T(  # SSPRK(5,4) in Butcher form
        (
            S(0, ()),
            S(0.391752226869254, (0.391752226869254,)),
            S(0.586079689066902, (0.217669096357835, 0.368410592709067)),
            S(0.474542363162481, (0.082692086683094, 0.139958502107426, 0.251891774371961)),
            S(0.935010631095793, (0.067966283574048, 0.115034698453668, 0.207034898772937, 0.54497475029514)),
        ),
        (0.146811876157876, 0.248482909391317, 0.104258830279481, 0.274438901048481, 0.226007483122845),
    )
    alphas = [[1], [0.444370493651235, 0.555629506348765], [0.620101851488403, 0, 0.379898148511597], [0.178079954393132, 0, 0, 0.821920045606868], [0, 0, 0.517231671970585, 0.096059710526147, 0.386708617503269]]
    betas = [[0.391752226571890], [0, 0.368410593050371], [0, 0, 0.251891774271694], [0, 0, 0, 0.544974750228521], [0, 0, 0, 0.063692468666290, 0.226007483236906]]
    assert distance(ssp45, B.from_shu_osher(alphas, betas).compose()) < 1e-8
    # serialisation round trips in every layout
    ck = PTab.RKE5.CashKarp.tableau()
    coeffs = B.decompose(T(ck.stages, ck.weights))
    n = len(coeffs.c)
    for compute_c, b_last in itertools.product((False, True), (True, False)):
        flat = ([] if compute_c else list(coeffs.c)) + ([] if b_last else list(coeffs.b)) + [v for row in coeffs.a[1:] for v in row] + (list(coeffs.b) if b_last else [])
        back = B.deserialize(flat, n, compute_c, b_last).compose()
        assert distance(back, T(ck.stages, ck.weights)) < 1e-15
    shifted = B.empty(3, one_index=True)
    assert len(shifted.c) == 4 and shifted.c[1] == 0 and [len(r) for r in shifted.a] == [0, 1, 2, 3]
    # no two built-in methods of the same size coincide (reference test_tableau_dupe), graveyard included
    everything = [m.tableau() for m in (*PTab.BUILTIN_TABLEAUX, *PTab.GRAVEYARD)]
    for i, a in enumerate(everything):
        for b in everything[i + 1 :]:
            if len(a.stages) == len(b.stages):
                assert distance(a, b) > 1e-2
    for provider in PF.STABLE_PROVIDERS.values():  # reference test_tableau_preset_nondefault
        assert provider not in PF.DEFAULT_PROVIDERS.values()


def test_rk_wrapper_trim_indices_name_the_stage_of_each_timestep():
    """`timesteps` of the Runge-Kutta wrappers is all_points WITHOUT the stages on the clean end; element k of it
    is stage trim_indices[k] of the walk.  Cash-Karp (RKUltra order 6) and SSPRK3 have a c = 1 stage ahead of their last one, so
    the last elements sit one position earlier than their stage (reference diffusers.py:652-666 builds the same trimmed table)."""
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.sampling import tableaux as TB

    seen_shift = 0
    for mk in (
        lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6),
        lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3, providers={3: TB.SSP.RK3_3}),
        lambda: PD.RKUltraWrapperScheduler(PS.Linear(), sampler_order=4),
        lambda: PD.DynasauRKWrapperScheduler(PS.Scaled(), sampler_order=6),
    ):
        for steps in (1, 3, 7):
            w = mk()
            w.set_timesteps(steps)
            kept = w.trim_indices
            assert len(kept) == len(w.schedule_np_trim) == len(w.timesteps)
            for k, at in enumerate(kept):
                assert w.schedule_np_trim[k, 0] == w.all_points[at].timestep
            assert list(kept) == sorted(set(kept))
            seen_shift += list(kept) != list(range(len(kept)))
            # the walk visits exactly the kept stages, in order, when fed the table (host floats)
            import torch

            x = torch.zeros(1, 2, 4, 4)
            visited = []
            for t in w.timesteps.tolist():
                visited.append(w._index)
                x = w.step(torch.ones_like(x), t, x, return_dict=False)[0]
            assert visited == list(kept)
    assert seen_shift >= 2  # the case the check used to get wrong does occur


def test_launch_hooks_are_per_thread():
    "ADVICE r2 (low): `_hip.trace` / `_hip.indexed` installed by one thread must not record or redirect another thread's launches"
    import threading

    from skrample_amd import _hip

    seen = {}
    _hip.trace = []
    _hip.indexed = "rows-of-this-thread"
    try:
        def other():
            seen["trace"], seen["indexed"] = _hip.trace, _hip.indexed
            _hip.trace = ["theirs"]

        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == {"trace": None, "indexed": None}
        assert _hip.trace == [] and _hip.indexed == "rows-of-this-thread"
    finally:
        _hip.trace = None
        _hip.indexed = None
    assert _hip.trace is None and _hip.indexed is None


def test_lazy_generator_list_behaves_like_the_list_it_replaces():
    "BatchTensorNoise.generators: one object per batch item as in the reference (noise.py:438-446), built on access"
    import dataclasses

    from skrample_amd.pytorch import noise as PN

    lazy = PN._LazyGenerators(PN.Random, (4, 8, 8), [5, 6, 7], None, torch.float32)
    assert len(lazy) == 3 and not lazy._made
    assert isinstance(lazy[1], PN.Random) and lazy[1].seed == 6 and lazy[-1].seed == 7 and lazy[1] is lazy[1]
    assert [g.seed for g in lazy] == [5, 6, 7] and [g.seed for g in lazy[1:]] == [6, 7]
    assert lazy == [PN.Random.from_inputs((4, 8, 8), s, dtype=torch.float32) for s in (5, 6, 7)]
    with pytest.raises(IndexError):
        lazy[3]
    assert dataclasses.is_dataclass(PN.BatchTensorNoise) and [f.name for f in dataclasses.fields(PN.BatchTensorNoise)][0] == "generators"
