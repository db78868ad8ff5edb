"""Round-4 device tests: the in-order replay fast path of SkrampleWrapperScheduler.step (library-side step programs,
skr_program_create / skr_program_launch) against the general path, bit for bit; its refusals; the alias guard inside it."""

import pytest
import torch

import skrample_amd.diffusers as PD
import skrample_amd.scheduling as PS
from skrample_amd import _hip
from skrample_amd.pytorch import noise as PN
from skrample_amd.sampling import lazy
from skrample_amd.sampling import models as PM
from skrample_amd.sampling import structured as PT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    _hip.load()
    return torch.device("cuda:0")


MAKERS = {
    # name: (factory, fast path expected to serve steps)
    "dpm2_sde_karras": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())), True),
    "dpm3_ode": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=3), PS.Scaled()), True),
    "euler_sde": (lambda: PD.SkrampleWrapperScheduler(PT.Euler(stochasticity=1), PS.Scaled()), True),
    "euler_ode_flow": (lambda: PD.SkrampleWrapperScheduler(PT.Euler(), PS.Linear(), PM.FlowModel()), True),
    "adams4_v_zsnr": (lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()), True),
    "unipc3_sde_flow": (lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel()), True),
    "unipc2_adams3_v": (lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=2, predictor=PT.Adams(order=3)), PS.Scaled(), PM.VelocityModel()), True),
    "spc": (lambda: PD.SkrampleWrapperScheduler(PT.SPC(), PS.Scaled()), True),
    "dpm2_alias_true": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), alias_history=True), True),
    "dpm2_snapshots": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), alias_history=False), False),
    "euler_inverted": (lambda: PD.SkrampleWrapperScheduler(PT.Euler(stochasticity=1), PS.Scaled(), invert_prediction=True), False),
    "dpm2_offset_noise": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Offset, noise_props=PN.OffsetProps()), False),
    "dpm2_fp64_scale": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), compute_scale=torch.float64), None),
}


def trajectory(w, x0, outs, seeds, as_float=True):
    w.set_timesteps(len(outs))
    ts = w.timesteps.tolist() if as_float else list(w.timesteps)
    x, traj = x0, []
    for i, t in enumerate(ts):
        prev, pred = w.step(outs[i], t, x, generator=seeds, return_dict=False)
        traj.append((prev, torch.as_tensor(pred.materialize() if isinstance(pred, lazy.LazyTensor) else pred)))
        x = prev
    return traj


@pytest.mark.parametrize("name", sorted(MAKERS))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fast_steps_equal_the_general_path(name, dtype, dev):
    mk, expect_fast = MAKERS[name]
    shape, steps = (3, 4, 32, 32), 9
    g = torch.Generator().manual_seed(41)
    x0 = torch.randn(shape, generator=g).to(dtype).to(dev)
    outs = [torch.randn(shape, generator=g).to(dtype).to(dev) for _ in range(steps)]
    seeds = [5, 6, 7]
    fast, general = mk(), mk()
    general.fast_steps = False
    want = trajectory(general, x0, outs, seeds)
    for rep in range(4):
        got = trajectory(fast, x0, outs, seeds)
        for i, (a, b) in enumerate(zip(got, want)):
            assert a[0].dtype == b[0].dtype and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (name, rep, i)
    assert general._fast_hits == 0
    if expect_fast is None:
        pass
    elif expect_fast:
        # from the second run on every step of an in-order run is served by the fast path ("auto" aliasing settles at the third call of a run)
        assert fast._fast_hits >= 3 * (steps - 2), (name, fast._fast_hits)
    else:
        assert fast._fast_hits == 0, (name, fast._fast_hits)
    # the return_dict form and 0-d tensor timesteps (general path) agree too
    assert torch.equal(trajectory(fast, x0, outs, seeds, as_float=False)[-1][0], want[-1][0])
    fast.set_timesteps(steps)
    d = fast.step(outs[0], fast.timesteps.tolist()[0], x0, generator=seeds)
    assert torch.equal(d.prev_sample, want[0][0]) and torch.equal(d["prev_sample"], want[0][0])


def test_fast_path_refusals_fall_back_to_the_general_path(dev):
    "another timestep than the next one, other shapes, a begin index: the general path serves them, with the same bits as a wrapper that never had a fast path"
    mk = MAKERS["dpm2_alias_true"][0]
    shape, steps = (2, 4, 32, 32), 8
    g = torch.Generator().manual_seed(43)
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)
    outs = [torch.randn(shape, generator=g).bfloat16().to(dev) for _ in range(steps)]
    fast, general = mk(), mk()
    general.fast_steps = False
    for _ in range(3):
        trajectory(fast, x0, outs, [1, 2])
    hits = fast._fast_hits
    assert hits > 0

    def odd_run(w):
        w.set_timesteps(steps)
        ts = w.timesteps.tolist()
        res = []
        x = x0
        for i in (0, 1, 3, 4, 2, 5):  # out of order from the third call on
            x = w.step(outs[i], ts[i], x, generator=[1, 2], return_dict=False)[0]
            res.append(x)
        return res

    for a, b in zip(odd_run(fast), odd_run(general)):
        assert torch.equal(a, b)
    assert fast._fast_hits == hits + 2  # the two in-order calls only

    def begin_run(w):  # an image-to-image style run: begin index 3
        w.set_timesteps(steps)
        w.set_begin_index(3)
        ts = w.timesteps.tolist()
        x, res = x0, []
        for i in range(3, steps):
            x = w.step(outs[i], ts[i], x, generator=[1, 2], return_dict=False)[0]
            res.append(x)
        return res

    want = begin_run(general)
    for rep in range(3):
        for a, b in zip(begin_run(fast), want):
            assert torch.equal(a, b)
    # another batch size: entries are per shape, the general path re-learns
    big = torch.cat([x0, x0])
    bouts = [torch.cat([o, o]) for o in outs]
    want = trajectory(general, big, bouts, [1, 2, 3, 4])
    for rep in range(3):
        for a, b in zip(trajectory(fast, big, bouts, [1, 2, 3, 4]), want):
            assert torch.equal(a[0], b[0])


@pytest.mark.parametrize("maker", ["dpm2_alias_true", "adams4_v_zsnr"])
def test_fast_path_checks_what_its_history_pointers_point_at(maker, dev):
    """a replayed step binds its history operands by remembered raw pointers.  A record appended by a general-path call of
    ANOTHER dtype (same run, one step handed over as fp32) must not be bound by the next step's entry, which was learned over bf16
    history: it is refused (general path, which re-lowers for the mixed history) and the bits are those of a wrapper without a fast path."""
    mk = MAKERS[maker][0]
    shape, steps = (2, 4, 32, 32), 8
    g = torch.Generator().manual_seed(47)
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)
    outs = [torch.randn(shape, generator=g).bfloat16().to(dev) for _ in range(steps)]
    fast, general = mk(), mk()
    general.fast_steps = False
    for _ in range(3):
        trajectory(fast, x0, outs, [1, 2])
    assert fast._fast_hits > 0

    def run(w, odd_at, odd):
        w.set_timesteps(steps)
        ts = w.timesteps.tolist()
        x, res = x0, []
        for i in range(steps):
            xi, oi = (odd(x), odd(outs[i])) if i == odd_at else (x, outs[i])
            x = w.step(oi, ts[i], xi, generator=[1, 2], return_dict=False)[0]
            x = x.to(torch.bfloat16) if x.dtype != torch.bfloat16 else x
            res.append(x)
        return res

    for odd_at in (2, 4):
        want = run(general, odd_at, lambda t: t.float())
        hits = fast._fast_hits
        got = run(fast, odd_at, lambda t: t.float())
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert 0 < fast._fast_hits - hits < steps - 1  # the odd step and the steps whose history holds its record went the general way
    # and a plain run afterwards is served by the entries again
    want = trajectory(general, x0, outs, [1, 2])
    hits = fast._fast_hits
    for a, b in zip(trajectory(fast, x0, outs, [1, 2]), want):
        assert torch.equal(a[0], b[0])
    assert fast._fast_hits - hits >= steps - 2


def test_alias_guard_inside_the_fast_path(dev):
    "a caller that starts reusing buffers once the fast path serves the steps still gets the error, never a wrong step"
    mk = MAKERS["dpm2_alias_true"][0]
    shape, steps = (2, 4, 16, 16), 6
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)
    outs = [torch.randn(shape, generator=g).bfloat16().to(dev) for _ in range(steps)]
    w = mk()
    for _ in range(3):
        trajectory(w, x0, outs, [1, 2])
    assert w._fast_hits > 0
    w.set_timesteps(steps)
    ts = w.timesteps.tolist()
    buf = torch.empty_like(outs[0])
    buf.copy_(outs[0])
    x = w.step(buf, ts[0], x0, generator=[1, 2], return_dict=False)[0]
    buf.copy_(outs[1])  # the network writes its next output into the buffer the history still reads
    with pytest.raises(_hip.SkrampleHipError, match="alias_history=False"):
        w.step(buf, ts[1], x, generator=[1, 2], return_dict=False)
    # static storage handed back as a new tensor object (no version bump)
    w.set_timesteps(steps)
    x = w.step(buf, ts[0], x0, generator=[1, 2], return_dict=False)[0]
    with pytest.raises(_hip.SkrampleHipError, match="model_output"):
        w.step(buf.view(shape), ts[1], x, generator=[1, 2], return_dict=False)


def test_seed_vectors_are_shared_between_runs_but_never_with_a_captured_loop(dev):
    from skrample_amd.graphs import capture_sampling_loop

    a = PN.seeds_tensor([11, 12, 13], dev)
    assert PN.seeds_tensor([11, 12, 13], dev) is a and PN.seeds_tensor([11, 12, 14], dev) is not a
    shape = (3, 4, 32, 32)
    g = torch.Generator(device=dev).manual_seed(3)
    x0 = torch.randn(shape, device=dev, generator=g).bfloat16()
    outs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(4)]
    n = [0]

    def net(x, t):
        n[0] += 1
        return outs[n[0] % 4]

    mk = MAKERS["dpm2_alias_true"][0]
    loop = capture_sampling_loop(mk(), net, x0, 5, seeds=[11, 12, 13])
    first = loop(x0).clone()
    loop(x0, seeds=[21, 22, 23])  # overwrites the loop's own seed buffer in place
    assert torch.equal(PN.seeds_tensor([11, 12, 13], dev).cpu(), torch.tensor([11, 12, 13]))  # a fresh upload, not the overwritten buffer
    assert torch.equal(loop(x0, seeds=[11, 12, 13]), first)


@pytest.mark.parametrize("unit", [(4, 5, 96, 96), (16, 21, 64, 64), (2, 8, 32, 32), (8, 3, 16, 48), (16, 32, 16, 16), (4, 7, 128, 128)])
def test_video_units_run_on_hand_written_kernels_only(unit, dev):
    """4-axis units (channels x frames x height x width) whose channel count is a power of two <= 16, with <= 32 frames and planes the
    LDS plane kernels take: planes on those kernels, BOTH outer axes in one fused pass (any_outer_two) -- no hipFFT plan is created and no
    hipFFT transform runs (skr_stat counters); against the oracle, and against the hipFFT route on the same seeds"""
    import os

    import numpy as np
    from conftest import note_margin
    from skr_oracle import noise as ON
    from skrample_amd.common import Step

    lib = _hip.load()
    seeds = [31, 32]

    def spec_normal(seed, stream, shape):
        return torch.from_numpy(ON.philox_normal(seed, stream, int(np.prod(shape)))).reshape(tuple(shape))

    plans, execs = lib.skr_stat(b"hipfft_plans"), lib.skr_stat(b"hipfft_execs")
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
    outs = []
    for n, st in enumerate((None, Step(0.45, 0.5))):
        got = g.generate(st).cpu()
        outs.append(got)
        ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), st) for s in seeds])
        exact = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape).double(), st) for s in seeds])
        err = ((got.double() - ref.double()).abs().max() / ref.double().abs().max()).item()
        note_margin("colored (4-axis units, plane kernels + fused outer axes)", "rel inf-norm error vs the fp32 oracle", err, 1e-5)
        far = ((got.double() - exact).abs().max() / exact.abs().max()).item()
        near = ((ref.double() - exact).abs().max() / exact.abs().max()).item()
        assert err < 1e-5 and far <= max(3.0 * near, 2e-6), (unit, st, err, far, near)  # (measured: 5.0e-7 at most, profiles/r04_parity_margins.txt)
    assert lib.skr_stat(b"hipfft_plans") == plans and lib.skr_stat(b"hipfft_execs") == execs, "a hipFFT plan was created / run for a shape the hand-written kernels cover"
    os.environ["SKR_FFT_NO_PLANES"] = "1"
    assert lib.skr_set_tuning(b"hipfft", 1) == 0  # (the N-D transform on hipFFT: the library's own any-length kernels are the default)
    try:
        h = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
        via_hipfft = [h.generate(st).cpu() for st in (None, Step(0.45, 0.5))]
    finally:
        del os.environ["SKR_FFT_NO_PLANES"]
        assert lib.skr_set_tuning(b"hipfft", -1) == 0
    assert lib.skr_stat(b"hipfft_execs") > execs  # (the comparison route did use it)
    for a, b in zip(outs, via_hipfft):
        assert ((a - b).abs().max() / b.abs().max()).item() < 1e-5


@pytest.mark.parametrize("config", ["headline", "cfg3c", "cfg5"])
def test_bench_line_of_a_config_on_the_device(config):
    """`python bench.py --config <c>` end to end on the device (short run, no CPU baseline / counter passes): one JSON line with the
    contract's keys, the config's own shape and 8(d) bytes, the step kernels' time from HIP events, and for the configs that name a
    noise generator the separate whole-step figures"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", config, "--steps", "12", "--warmup", "3", "--precondition", "12", "--sets", "4",
                          "--no-cpu-baseline", "--no-traffic", "--no-extras"], capture_output=True, text=True, timeout=600)  # fmt: skip
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-500:]  # ONE line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["steps"] == 12 and d["warmup"] == 3 and d["n_gpus"] == 1 and d["config"]["name"] == config and d["unit"] == "steps/s"
    per_elem = {"headline": 10, "cfg3c": 30, "cfg5": 100}[config]
    r = d["roofline"]
    assert r["algorithmic_bytes_per_element"] == per_elem and r["bound"] == "hbm" and 0.2 < r["frac"] < 1.0
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert ("whole_step" in r) == (config != "headline")
    if config != "headline":
        assert r["whole_step"]["us_per_step"] > r["us_per_step"] and r["whole_step"]["generator_us_per_step"] > 0


def test_two_threads_on_two_streams_do_not_disturb_each_other(dev):
    """serving: two Python threads, each with its own stream, scheduler and noise generators (Colored / Pyramid / in-kernel Philox), stepping
    at the same time -- every thread's trajectory equals the one it produces alone, bit for bit (workspaces, tickets, plan caches and the
    seed-vector cache are per generator / per thread; the library's tables are built once)"""
    import threading

    from skrample_amd.pytorch import noise as PN

    shape, steps = (4, 4, 64, 64), 6
    makers = [
        lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), noise_type=PN.Colored, noise_props=PN.ColoredProps()),
        lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Pyramid, noise_props=PN.PyramidProps()),
        lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3, stochasticity=1),
        lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Colored, noise_props=PN.ColoredProps()),
    ]
    g = torch.Generator().manual_seed(31)
    x0 = torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)
    outs = [torch.randn(shape, generator=g).to(torch.bfloat16).to(dev) for _ in range(steps * 3)]

    def run(k, rounds, sink, stream=None):
        try:
            with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev)):
                res = []
                for _ in range(rounds):
                    w = makers[k]()
                    w.set_timesteps(steps)
                    x = x0
                    for i, t in enumerate(w.timesteps):
                        x = w.step(outs[i], t, x, generator=[torch.Generator().manual_seed(100 + k * 10 + b) for b in range(shape[0])], return_dict=False)[0]
                    res.append(x.clone())
                torch.cuda.current_stream(dev).synchronize()
                sink[k] = res
        except BaseException as exc:  # noqa: BLE001
            sink[k] = exc

    alone: dict = {}
    for k in range(len(makers)):
        run(k, 1, alone)
        assert not isinstance(alone[k], BaseException), alone[k]
    torch.cuda.synchronize(dev)
    together: dict = {}
    threads = [threading.Thread(target=run, args=(k, 4, together, torch.cuda.Stream(device=dev))) for k in range(len(makers))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive()
    for k in range(len(makers)):
        assert not isinstance(together[k], BaseException), together[k]
        for res in together[k]:
            assert torch.equal(res, alone[k][0]), k
