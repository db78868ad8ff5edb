"""Host-resident operands (the reference's generic `T`: CPU torch tensors and numpy arrays, common.py:11-17; BASELINE
config 1 is "on CPU torch").  They are evaluated by skrample_amd's own host executor (sampling/lazy.py::_host_evaluate):
the same collapsed linear forms the HIP kernel runs, in plain torch.  These tests need no GPU: they replay the
reference-recorded fixtures and the oracle on the CPU, and check that device work can never end up here."""

import zlib

import numpy as np
import pytest
import torch
from cases import MODELS, NATIVE16_TAGS, SAMPLERS, SCHEDULES, oracle_schedule
from conftest import load_npz
from test_step_gpu import (
    EXTRA2_WRAPPERS,
    EXTRA3_WRAPPERS,
    EXTRA4_WRAPPERS,
    EXTRA_WRAPPERS,
    FIXTURE_WRAPPERS,
    NATIVE_SWEEP_COUNT,
    SWEEP_COUNT,
    SWEEP_NAMES,
    Injected,
    assert_close,
    native16_engine_vs_reference,
    replay_fixture,
    replay_native_api16,
    replay_native_sweep,
    sweep_case,
)

import skrample_amd.diffusers as PD
import skrample_amd.scheduling as PS
from skr_oracle import wrapper as OW
from skrample_amd import _hip
from skrample_amd.common import Step
from skrample_amd.sampling import lazy
from skrample_amd.sampling import models as PM
from skrample_amd.sampling import structured as PT

CPU = torch.device("cpu")
pytestmark = []  # (test_step_gpu's module-level gpu mark does not apply here)


@pytest.mark.parametrize("name", FIXTURE_WRAPPERS)
def test_baseline_config_fixtures_on_cpu(name):
    "the five BASELINE.json configs (reduced shape) on CPU tensors vs outputs of the reference itself; cfg1 is the reference's own CPU case"
    mk, dt = FIXTURE_WRAPPERS[name]
    replay_fixture(mk(), load_npz(f"steps_{name}.npz"), dt, CPU, name)


@pytest.mark.parametrize("name", EXTRA_WRAPPERS)
def test_extra_fixtures_on_cpu(name):
    blob = load_npz("steps_extra.npz")
    fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith(name + "/")}
    mk, dt = EXTRA_WRAPPERS[name]
    replay_fixture(mk(), fx, dt, CPU, name)


@pytest.mark.parametrize("name", EXTRA2_WRAPPERS)
def test_extra2_fixtures_on_cpu(name):
    blob = load_npz("steps_extra2.npz")
    fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith(name + "/")}
    mk, dt = EXTRA2_WRAPPERS[name]
    replay_fixture(mk(), fx, dt, CPU, name)


@pytest.mark.parametrize("name", EXTRA3_WRAPPERS)
def test_high_order_fixtures_on_cpu(name):
    blob = load_npz("steps_extra3.npz")
    fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith(name + "/")}
    mk, dt = EXTRA3_WRAPPERS[name]
    replay_fixture(mk(), fx, dt, CPU, name)


@pytest.mark.parametrize("name", EXTRA4_WRAPPERS)
def test_float64_compute_scale_on_16_bit_latents_on_cpu(name):
    blob = load_npz("steps_extra4.npz")
    fx = {k.split("/", 1)[1]: v for k, v in blob.items() if k.startswith(name + "/")}
    mk, dt = EXTRA4_WRAPPERS[name]
    replay_fixture(mk(), fx, dt, CPU, name)


@pytest.mark.parametrize("index", range(SWEEP_COUNT))
def test_reference_recorded_random_sweep_on_cpu(index):
    "64 seeded random wrapper configurations recorded from the reference itself, replayed on host tensors"
    m, fx, dt = sweep_case(index)
    replay_fixture(eval(m["text"], SWEEP_NAMES), fx, dt, CPU, m["text"], steps=m["steps"])


@pytest.mark.parametrize("index", range(NATIVE_SWEEP_COUNT))
def test_reference_recorded_sweep_without_a_compute_scale_on_cpu(index):
    "compute_scale=None on 16-bit host tensors: the wrappers (Runge-Kutta ones included) return the reference's bits"
    replay_native_sweep(index, CPU)


def test_reference_recorded_functional_loops_and_transforms_on_16_bit_host_tensors():
    replay_native_api16(CPU)


def test_configurations_the_reference_refuses_are_refused_here_too():
    "the sweep's rejects: set_timesteps / step raised in the reference -- the same configuration raises here (never a silent result)"
    import json

    from conftest import load_npz as load

    refused = json.loads(str(load("steps_sweep.npz")["refused"]))
    assert refused
    for r in refused:
        dt = getattr(torch, r["dtype"])
        with pytest.raises((ZeroDivisionError, ValueError, AssertionError, IndexError, AttributeError, TypeError, _hip.SkrampleHipError)):
            w = eval(r["text"], SWEEP_NAMES)
            w.set_timesteps(r["steps"])
            x = torch.zeros(r["shape"], dtype=dt)
            for t in w.timesteps:
                x = torch.as_tensor(w.step(torch.ones_like(x), t, x, return_dict=False)[0])
            assert torch.isfinite(x.float()).all()


@pytest.mark.parametrize("tag", NATIVE16_TAGS)
def test_sampler_level_api_vs_reference_recorded_16_bit_runs_on_cpu(tag):
    "the fused form (native.mode = 'never'): within half a unit of the exact result, where the reference's chain is 4-47 units away"
    from skrample_amd.sampling import native

    before, native.mode = native.mode, "never"
    try:
        native16_engine_vs_reference(tag, CPU)
    finally:
        native.mode = before


@pytest.mark.parametrize("sampler", ["euler", "dpm2_sde", "adams4", "unipc3", "spc"])
def test_samplers_vs_oracle_on_cpu(sampler):
    if sampler not in SAMPLERS:
        pytest.skip(f"no sampler case named {sampler}")
    mk_o, mk_p = SAMPLERS[sampler]
    steps, shape = 7, (2, 3, 10, 6)
    for sname, mname in (("karras_scaled", "eps"), ("linear", "flow")):
        g = torch.Generator().manual_seed(zlib.crc32(f"{sampler}/{sname}".encode()))  # (str hashes change from process to process)
        w = PD.SkrampleWrapperScheduler(mk_p(), SCHEDULES[sname][1](), MODELS[mname][1])
        o = OW.StepDriver(mk_o(), oracle_schedule(sname, steps), MODELS[mname][0])
        w.set_timesteps(steps)
        o.set_timesteps(steps)
        noises = [torch.randn(shape, generator=g) for _ in range(steps)]
        w._noise_generator = Injected(noises, CPU)
        x = torch.randn(shape, generator=g)
        for i, t in enumerate(w.timesteps):
            out = torch.randn(shape, generator=g)
            got = w.step(out, t, x, return_dict=False)[0]
            ref = o.step(out, t, x, noise=noises[i])[0]
            assert_close(got, ref, torch.float32, f"{sampler}/{sname}/{mname} step {i}")
            x = ref


def test_numpy_arrays_through_the_sampler_protocol():
    "ndarray in -> ndarray out (reference tests/self_sampling.py:184-224), same numbers as CPU tensors"
    rng = np.random.default_rng(5)
    sampler, model, schedule = PT.DPM(order=2), PM.NoiseModel(), PS.Scaled()
    steps = 6
    xa = rng.standard_normal((3, 8)).astype(np.float64)
    xt = torch.from_numpy(xa.copy())
    prev_a, prev_t = [], []
    for i in range(steps):
        out = rng.standard_normal((3, 8))
        ra = sampler.sample(xa, out, Step.from_int(i, steps), model, schedule, None, prev_a)
        rt = sampler.sample(xt, torch.from_numpy(out.copy()), Step.from_int(i, steps), model, schedule, None, prev_t)
        assert isinstance(ra.final, np.ndarray) and ra.final.dtype == np.float64
        assert isinstance(rt.final, torch.Tensor) and rt.final.device.type == "cpu"
        np.testing.assert_array_equal(ra.final, rt.final.numpy())
        prev_a.append(ra), prev_t.append(rt)
        xa, xt = ra.final, rt.final
    assert np.isfinite(xa).all()


def test_numpy_arrays_keep_the_dtypes_the_reference_returns():
    """found by tools/sweep_vs_reference.py `array`: UniPC kept its state and prediction in float32 for float64 arrays; SPC's signed-power blend took
    no ndarrays at all -- and the reference's spowf (common.py:187-190) multiplies by an int64 sign array, so numpy promotes a float32 array to
    float64 from the blended sample on, while the converted prediction of a DataModel stays what the network output was"""
    rng = np.random.default_rng(0)
    for dt in (np.float64, np.float32):
        x, out = rng.standard_normal((2, 3, 4)).astype(dt), rng.standard_normal((2, 3, 4)).astype(dt)
        for sampler in (PT.UniPC(order=2), PT.UniPC(order=3, stochasticity=1), PT.SPC()):
            prev, xx = [], x
            for i in range(4):
                rec = sampler.sample(xx, out, Step.from_int(i, 6), PM.NoiseModel(), PS.Scaled(), rng.standard_normal((2, 3, 4)).astype(dt) if sampler.require_noise else None, tuple(prev))
                pred = rec.prediction.materialize() if isinstance(rec.prediction, lazy.LazyTensor) else rec.prediction
                assert isinstance(rec.final, np.ndarray) and rec.final.dtype == dt and np.asarray(pred).dtype == dt and np.asarray(rec.sample).dtype == dt, (sampler, i)
                prev.append(rec)
                xx = rec.final
    x, out = rng.standard_normal((2, 3, 4)).astype(np.float32), rng.standard_normal((2, 3, 4)).astype(np.float32)
    sampler, prev, xx = PT.SPC(power=0.5), [], x
    for i in range(3):
        rec = sampler.sample(xx, out, Step.from_int(i, 6), PM.DataModel(), PS.Scaled(), None, tuple(prev))
        pred = rec.prediction.materialize() if isinstance(rec.prediction, lazy.LazyTensor) else rec.prediction
        assert rec.final.dtype == (np.float32 if i == 0 else np.float64) and np.asarray(pred).dtype == np.float32, i
        prev.append(rec)
        xx = rec.final


def test_adaptive_runge_kutta_takes_ndarrays():
    "found by the live sweep: RKMoire's embedded pair asked the evaluator for a numpy dtype, and its error norm refused ndarrays"
    import skrample_amd.sampling.functional as PF

    rng = np.random.default_rng(3)
    toy = lambda xx, t, s, a: xx * 0.3 - 0.1 * s + 0.05 * a  # noqa: E731
    for dt in (np.float64, np.float32):
        x = rng.standard_normal((2, 3)).astype(dt)
        for sampler in (PF.RKMoire(order=2), PF.RKMoire(order=3, evaluator=PF.FunctionalAdaptive.mae), PF.RKUltra(order=4), PF.DynasauRK(order=3)):
            res = sampler.sample_model(x.copy(), toy, PM.FlowModel(), PS.Linear(), 6)
            assert isinstance(res, np.ndarray) and res.dtype == dt and np.isfinite(res).all(), sampler
            twin = sampler.sample_model(torch.from_numpy(x.copy()), toy, PM.FlowModel(), PS.Linear(), 6)
            np.testing.assert_allclose(res, twin.numpy(), rtol=1e-6 if dt == np.float32 else 1e-12, atol=1e-7 if dt == np.float32 else 1e-13)


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_a_scheduler_copied_in_the_middle_of_a_run_goes_on_like_the_original(device):
    """copy.deepcopy / pickle of a wrapper after some steps (a plain dataclass in the reference: either works there): the copy's history guard is
    stamped with the copy's own tensors, pointer-bound replay state is dropped, and copy and original continue to the same result"""
    import copy
    import pickle

    from skrample_amd.pytorch import noise as PN

    makers = [
        lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())),
        lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3), PS.Linear(), PM.FlowModel()),
        lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.Scaled(), alias_history=True),
    ]
    if device == "cuda":
        makers.append(lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Colored, noise_props=PN.ColoredProps()))
        makers.append(lambda: PD.SkrampleWrapperScheduler(PT.Euler(stochasticity=1), PS.Scaled(), noise_type=PN.Pyramid, noise_props=PN.PyramidProps()))
    for mk in makers:
        w = mk()
        w.set_timesteps(8)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(2, 4, 16, 16, generator=g).to(device)
        outs = [torch.randn(2, 4, 16, 16, generator=g).to(device) for _ in range(8)]
        seeds = [11, 12] if device == "cuda" else [torch.Generator().manual_seed(i) for i in range(2)]
        for i, t in enumerate(w.timesteps[:4]):
            x = torch.as_tensor(w.step(outs[i], t, x, generator=seeds, return_dict=False)[0])
        if device == "cuda":
            torch.cuda.synchronize()
        copies = [copy.deepcopy(w), pickle.loads(pickle.dumps(w))] if device == "cpu" else [copy.deepcopy(w)]
        ends = []
        for runner in (*copies, w):
            xx = x
            for i, t in enumerate(runner.timesteps[4:], start=4):
                xx = torch.as_tensor(runner.step(outs[i], t, xx, return_dict=False)[0])
            ends.append(xx)
        for other in ends[:-1]:
            assert torch.equal(other, ends[-1]), type(w.sampler).__name__


def test_wrapper_random_noise_on_cpu_uses_the_callers_generators():
    "host-resident latents draw white noise as the reference does: torch.randn from one CPU generator per sample"
    shape, steps = (2, 4, 8, 8), 4
    mk = lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()))  # noqa: E731
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(shape, generator=g)
    outs = [torch.randn(shape, generator=g) for _ in range(steps)]

    def run(seed_base):
        w = mk()
        w.set_timesteps(steps)
        gens = [torch.Generator().manual_seed(seed_base + i) for i in range(shape[0])]
        x = x0
        for out, t in zip(outs, w.timesteps):
            x = w.step(out, t, x, generator=gens, return_dict=False)[0]
        return x

    a, b, c = run(10), run(10), run(11)
    assert torch.equal(a, b) and not torch.equal(a, c) and a.dtype == torch.float32 and a.device.type == "cpu"
    # reproduce the first step's noise by hand from the same generators
    w = mk()
    w.set_timesteps(steps)
    gens = [torch.Generator().manual_seed(10 + i) for i in range(shape[0])]
    expect = torch.stack([torch.randn(shape[1:], generator=torch.Generator().manual_seed(10 + i)) for i in range(shape[0])])
    w.step(outs[0], w.timesteps[0], x0, generator=gens, return_dict=False)
    assert w._noise_generator._draws == 1
    got = torch.stack([torch.randn(shape[1:], generator=torch.Generator().manual_seed(10 + i)) for i in range(shape[0])])
    assert torch.equal(expect, got)
    from skrample_amd.pytorch import noise as PN

    # structured generators on host tensors run on the host generators too (tests/test_host_noise.py); only Brownian has no host form
    w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Brownian)
    w.set_timesteps(steps)
    with pytest.raises(_hip.SkrampleHipError, match="torchsde"):
        w.step(outs[0], w.timesteps[0], x0, generator=gens, return_dict=False)


def test_device_work_never_reaches_the_host_executor(monkeypatch):
    """the host executor serves host-resident operands only: a form over (fake) device tensors goes to the C ABI and fails
    loudly when the library is missing; mixed residency is refused"""
    calls = []
    monkeypatch.setattr(lazy, "_host_evaluate", lambda *a, **k: calls.append(a) or [torch.zeros(1)])
    cpu_form = lazy.Lin.leaf(torch.ones(8)) * 2.0
    lazy.evaluate([cpu_form], [torch.float32])
    assert len(calls) == 1

    class FakeDevice(torch.Tensor):  # a tensor that reports a HIP device without one being present
        @property
        def device(self):
            return torch.device("cuda", 0)

        @property
        def is_cuda(self):
            return True

    fake = torch.ones(8).as_subclass(FakeDevice)
    form = lazy.Lin({id(fake): (fake, 1.0)}, fake.shape, torch.device("cuda", 0))
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libskrample_hip.so")
    monkeypatch.setattr(_hip, "_lib", None)
    with pytest.raises(Exception) as err:  # SkrampleHipError from the loader (or torch refusing the fake device first)
        lazy.evaluate([form], [torch.float32])
    assert len(calls) == 1, "a device form must not be routed to the host executor"
    assert "libskrample_hip" in str(err.value) or "cuda" in str(err.value).lower() or "hip" in str(err.value).lower()


def test_mixed_residency_is_refused():
    a = lazy.Lin.leaf(torch.ones(8))

    class FakeDevice(torch.Tensor):
        @property
        def device(self):
            return torch.device("cuda", 0)

    fake = torch.ones(8).as_subclass(FakeDevice)
    mixed = lazy.Lin({**a.terms, id(fake): (fake, 1.0)}, a.shape, a.device)
    with pytest.raises(_hip.SkrampleHipError, match="all live on the HIP device or all on the host"):
        lazy.evaluate([mixed], [torch.float32])


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_empty_batch_through_the_wrappers(device):
    "B = 0: every wrapper steps an empty batch to an empty batch (no generator to seed, no buffer for the alias guard to compare)"
    makers = [
        lambda: PD.SkrampleWrapperScheduler(PT.Euler(), PS.Scaled()),
        lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())),
        lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Scaled()),
        lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3, stochasticity=1),
    ]
    for mk in makers:
        w = mk()
        w.set_timesteps(3)
        x = torch.zeros(0, 4, 8, 8, dtype=torch.bfloat16, device=device)
        for t in w.timesteps:
            x = torch.as_tensor(w.step(torch.zeros_like(x), t, x, return_dict=False)[0])
            assert tuple(x.shape) == (0, 4, 8, 8) and x.dtype == torch.bfloat16 and x.device.type == device
