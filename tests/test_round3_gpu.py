"""Round 3: the compositions and plumbing the round-2 review found untested on the device.

* BASELINE config 5 as specified -- RKUltra-6 SDE + **Pyramid** noise through `RKUltraWrapperScheduler` -- against the
  oracle's inside-out Runge-Kutta driver fed the same realised Pyramid tensors (reference diffusers.py:746-873 with
  pytorch/noise.py:125-207), at a small shape and on three samples of the full 64x4x256x256 shard;
* BASELINE config 3 as specified -- UniPC-3 SDE + **Colored** noise -- at the full 256x16x128x128 size (three oracle samples):
  the launch with two noise-tensor operands (10 + 1 operands, two outputs) at the size it runs in production;
* whole sampling loops driven by DEVICE timesteps (`for t in scheduler.timesteps`) for tableaux that have a c = 1 stage ahead
  of their last one (the trimmed table is shorter than the stage list);
* the batch-shard path in two separate processes, each started with torchrun-style environment variables and running its
  `BatchShard.from_env` slice on the device: the concatenation equals the single-process run bit for bit.
"""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT
from test_step_gpu import assert_close

import skrample_amd.diffusers as PD
import skrample_amd.scheduling as PS
from skr_oracle import rk as OK
from skr_oracle import samplers as OA
from skr_oracle import schedules as OS
from skr_oracle import wrapper as OW
from skrample_amd import _hip
from skrample_amd.common import Step
from skrample_amd.pytorch import noise as PN
from skrample_amd.sampling import models as PM
from skrample_amd.sampling import structured as PT
from skrample_amd.sampling import tableaux as TB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    _hip.load()
    return torch.device("cuda:0")


def cfg5_wrapper():
    return PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps())


def cfg3_wrapper():
    return PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps())


def test_cfg5_rkultra6_pyramid_vs_oracle_small(dev):
    "free-running: the oracle's trajectory is fed forward, the wrapper is teacher-forced with it (every stage of 3 steps)"
    steps, shape, seeds = 3, (2, 4, 32, 32), [71, 72]
    w = cfg5_wrapper()
    o = OW.RKDriver(OK.pick_tableau(6), OS.scaled(), "eps", "data", 1.0)
    w.set_timesteps(steps)
    o.set_timesteps(steps)
    np.testing.assert_allclose(w.timesteps.numpy(), o.timesteps.numpy(), rtol=0, atol=1e-9)
    shadow = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, shape[1:], seeds, props=PN.PyramidProps(), dtype=torch.bfloat16)  # generator dtype = sample dtype (diffusers.py:343)
    drawn = []

    def noise_fn(step=None):  # the draw the wrapper's own generator makes for this step (draw n <-> step n)
        drawn.append(1)
        return shadow.generate(step).float().cpu()  # the wrapper widens the drawn tensor to compute_scale (diffusers.py:343-346)

    g = torch.Generator().manual_seed(17)
    x = torch.randn(shape, generator=g).bfloat16()
    for i, t in enumerate(w.timesteps):
        out = (torch.randn(shape, generator=g) * 0.5 + x.float() * 0.3).bfloat16()
        got = w.step(out.to(dev), t, x.to(dev), generator=seeds, return_dict=False)[0]
        ref = o.step(out, o.timesteps[i], x, noise_fn=noise_fn)
        assert_close(got, ref, torch.bfloat16, f"stage call {i}", flips=0.10)
        x = ref
    assert len(drawn) == steps  # one draw per step, at its last stage


FULL = {
    # name: (wrapper, oracle, noise kind, props, per-GPU shape, calls)
    "cfg5_rkultra6_sde_pyramid": (cfg5_wrapper, lambda: OW.RKDriver(OK.pick_tableau(6), OS.scaled(), "eps", "data", 1.0), PN.Pyramid, PN.PyramidProps(), (64, 4, 256, 256), 12),
    "cfg3_unipc3_sde_flow_colored": (cfg3_wrapper, lambda: OW.StepDriver(OA.make("unipc", 3, eta=1), OS.linear(), "flow"), PN.Colored, PN.ColoredProps(), (256, 16, 128, 128), 5),
}


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_size_configs_with_their_noise(name, dev):
    """The BASELINE configs that name a noise generator, at their full per-GPU size and WITH that generator: determinism, and the
    oracle (teacher-forced) on three samples, fed the tensors a three-sample generator with the same seeds realises -- a sample's
    noise depends on nothing but its own seed (tests/test_noise_gpu.py::test_generators_full_size_properties)."""
    mk_w, mk_o, kind, props, shape, calls = FULL[name]
    B, steps = shape[0], 20
    gd = torch.Generator(device=dev).manual_seed(977)
    x = torch.randn(shape, device=dev, generator=gd).bfloat16()
    outs = [torch.randn(shape, device=dev, generator=gd).bfloat16() for _ in range(2)]
    seeds = [4200 + i for i in range(B)]
    idx = [0, B // 2, B - 1]

    def run():
        w = mk_w()
        w.set_timesteps(steps)
        cur, ins, res = x, [], []
        for i in range(calls):
            o_ = (outs[i % 2] * 0.25 + cur * 0.5).bfloat16()
            ins.append((cur[idx].cpu(), o_[idx].cpu()))
            cur = w.step(o_, w.timesteps[i], cur, generator=seeds, return_dict=False)[0]
            res.append(cur[idx].cpu())
            del o_
        return ins, res, w

    ins, full, w = run()
    _, again, _ = run()
    for f, a in zip(full, again):
        assert torch.equal(f, a) and torch.isfinite(f.float()).all()
    torch.cuda.empty_cache()

    o = mk_o()
    o.set_timesteps(steps)
    np.testing.assert_allclose(w.timesteps.numpy()[:calls], o.timesteps.numpy()[:calls], rtol=0, atol=1e-9)
    shadow = PN.BatchTensorNoise.from_batch_inputs(kind, shape[1:], [seeds[j] for j in idx], props=props, dtype=torch.bfloat16)  # generator dtype = sample dtype (diffusers.py:343)
    for i in range(calls):
        xin, oin = ins[i]
        if isinstance(o, OW.RKDriver):
            ref = o.step(oin, o.timesteps[i], xin, noise_fn=lambda step=None: shadow.generate(step).float().cpu())
        else:
            ref = o.step(oin, o.timesteps[i], xin, noise=shadow.generate(Step.from_int(i, steps)).cpu())[0]
        assert_close(full[i], ref, torch.bfloat16, f"{name} call {i}", flips=0.10)


DEVICE_TIMESTEP_LOOPS = {
    "rkultra6_cashkarp": lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6),
    "rkultra6_cashkarp_sde": lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1),
    "dynasaurk6": lambda: PD.DynasauRKWrapperScheduler(PS.Scaled(), sampler_order=6),
    "ssp_rk3_3": lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3, providers={3: TB.SSP.RK3_3}),
    "rkultra5": lambda: PD.RKUltraWrapperScheduler(PS.Linear(), sampler_order=5, model=PM.FlowModel()),
    "rkultra2": lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=2),
}


@pytest.mark.parametrize("name", sorted(DEVICE_TIMESTEP_LOOPS))
def test_whole_loop_with_device_timesteps(name, dev):
    """`set_timesteps(n, device='cuda'); for t in sched.timesteps: sched.step(out, t, x)` -- the standard
    diffusers loop.  `timesteps` is the table WITHOUT the stages on the clean end, so its element k is not stage k of the tableau
    walk once a trimmed stage lies before it (Cash-Karp, Fehlberg, SSPRK3: a c = 1 stage ahead of the last one).  The loop must
    run to the end and equal the same loop driven by host floats, bit for bit."""
    mk = DEVICE_TIMESTEP_LOOPS[name]
    shape, steps = (2, 4, 16, 16), 3
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(shape, generator=g).to(dev)
    net = [torch.randn(shape, generator=g).to(dev) for _ in range(64)]

    def run(kind):
        w = mk()
        w.set_timesteps(steps, device=dev)
        ts = w.timesteps
        assert ts.is_cuda
        x, calls = x0, 0
        for k, t in enumerate(ts if kind == "device" else ts.tolist()):
            x = w.step(net[k] * 0.3 + x * 0.2, t, x, generator=[9, 10], return_dict=False)[0]
            calls += 1
        return x, calls

    ref, n_host = run("host")
    got, n_dev = run("device")
    assert n_host == n_dev == len(mk_table(mk, steps, dev))
    assert torch.equal(got, ref)
    # the table really is shorter than the stage list for the tableaux this test is about
    w = mk()
    w.set_timesteps(steps, device=dev)
    if name.startswith(("rkultra6", "ssp")):
        assert len(w.trim_indices) < len(w.all_points), "expected a trimmed clean-end stage in this tableau"
        assert list(w.trim_indices) != list(range(len(w.trim_indices)))
    # an element that names the wrong stage is still refused
    with pytest.raises(AssertionError):
        w.step(net[0], w.timesteps[1], x0)


def mk_table(mk, steps, dev):
    w = mk()
    w.set_timesteps(steps, device=dev)
    return w.timesteps


def test_issued_timesteps_belong_to_their_schedule(dev):
    "ADVICE r2 (low): an element of a `timesteps` tensor handed out for an EARLIER schedule no longer names a step of the new one"
    w = PD.SkrampleWrapperScheduler(PT.Euler(), PS.Scaled())
    x = torch.randn(2, 4, 16, 16, device=dev)
    w.set_timesteps(8, device=dev)
    old = w.timesteps
    w.set_timesteps(5, device=dev)
    new = w.timesteps
    y = w.step(x, new[0], x, return_dict=False)[0]
    assert torch.isfinite(y).all()
    w.set_timesteps(5, device=dev)
    _ = w.timesteps
    with pytest.raises(ValueError):  # 8-step table value 2 is not in the 5-step table: read back once, then list.index raises as in the reference
        w.step(x, old[2], x)
    # a foreign device scalar is read back once per tensor object, then served from the cache
    w.set_timesteps(5, device=dev)
    t0 = w.timesteps[0].clone()
    w.step(x, t0, x)
    assert id(t0) in w._foreign_timesteps
    w.set_timesteps(5, device=dev)
    assert w._foreign_timesteps == {}


def test_user_capture_after_a_run_that_drew_noise_ahead(dev):
    """an eager run with Pyramid noise leaves a side-stream event behind; a plain `torch.cuda.graph`
    capture of the next run (after reset_run / set_timesteps, no skrample_amd.graphs helper) must work and replay the eager
    result."""
    shape, steps, seeds = (2, 4, 32, 32), 4, [3, 4]
    w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Pyramid, noise_props=PN.PyramidProps(), prefetch_noise=True)
    w.set_timesteps(steps)
    ts = w.timesteps.tolist()
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)
    net = [torch.randn(shape, generator=g).bfloat16().to(dev) for _ in range(steps)]

    def loop(x):
        for k, t in enumerate(ts):
            x = w.step((net[k] * 0.5 + x * 0.25), t, x, generator=seeds, return_dict=False)[0]
        return x

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        eager = loop(x0)
        w.reset_run()
        eager2 = loop(x0)  # (warm: programs lowered, workspaces allocated)
    torch.cuda.current_stream(dev).wait_stream(side)
    assert torch.equal(eager, eager2)
    w.reset_run()  # run boundary: drains whatever the side stream still drew
    static_in = x0.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out = loop(static_in)
    graph.replay()
    torch.cuda.synchronize(dev)
    assert torch.equal(static_out, eager)


def test_captured_loop_refuses_a_slot_that_was_never_loaded(dev):
    "ADVICE r2 (low): replaying a table slot that no schedule was loaded into ran every step with all-zero scalars"
    from skrample_amd.graphs import capture_sampling_loop

    shape, steps, seeds = (2, 4, 32, 32), 4, [3, 4]
    mk = lambda **kw: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), **kw)  # noqa: E731
    net = lambda x, t: x * 0.5  # noqa: E731
    x0 = torch.randn(shape, device=dev).bfloat16()
    loop = capture_sampling_loop(mk(), net, x0, steps, seeds=seeds, indexed=True, slots=3)
    ref = loop(x0)
    assert torch.isfinite(ref.float()).all() and ref.float().abs().max() > 0
    with pytest.raises(ValueError, match="never been loaded"):
        loop(x0, slot=2)
    loop.retarget(mk(), slot=2)
    assert torch.equal(loop(x0, slot=2), ref)
    assert torch.equal(loop(x0, slot=0), ref)


@pytest.mark.parametrize("kind", ["dpm2_sde", "rk4_sde"])
def test_retargeted_loop_feeds_the_network_the_new_timesteps(kind, dev):
    """An indexed capture hands the network 0-d elements of one device-resident `timesteps` tensor (as `for t in scheduler.timesteps`
    does in a diffusers pipeline), and re-targeting / switching slots rewrites that tensor: a network that USES t follows the new
    schedule.  (With host floats the first schedule's timesteps would be frozen into the captured network.)"""
    from skrample_amd.graphs import capture_sampling_loop

    shape, steps, seeds = (3, 4, 32, 32), 6, [5, 6, 7]
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)
    net = lambda x, t: x * (0.4 + t.to(torch.float32) / 2500).to(x.dtype) + 0.1 * x.abs()  # noqa: E731  (t: 0-d device tensor)
    if kind == "dpm2_sde":
        mk = lambda sch: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), sch)  # noqa: E731
    else:
        mk = lambda sch: PD.RKUltraWrapperScheduler(sch, sampler_order=4, stochasticity=1)  # noqa: E731
    variants = [PS.Karras(PS.Scaled()), PS.Scaled(), PS.Exponential(PS.Scaled())]

    def eager(w, x):
        w.set_timesteps(steps, device=dev)
        ts = w.timesteps
        for i in range(ts.numel()):
            x = w.step(net(x, ts[i]), ts[i], x, generator=seeds, return_dict=False)[0]
        return x

    want = [eager(mk(sch), x0) for sch in variants]
    assert not torch.equal(want[0], want[1]) and not torch.equal(want[1], want[2])
    loop = capture_sampling_loop(mk(variants[0]), net, x0, steps, seeds=seeds, indexed=True, slots=3)
    assert loop.static_times is not None
    assert torch.equal(loop(x0), want[0])
    loop.retarget(mk(variants[1]), slot=1)
    loop.retarget(mk(variants[2]), slot=2)
    for k in (1, 2, 0, 2, 1):
        assert torch.equal(loop(x0, slot=k), want[k]), (kind, k)
    loop.retarget(mk(variants[1]), slot=0)  # the slot in use: its timesteps are replaced at once
    assert torch.equal(loop(x0, slot=0), want[1])
    assert torch.equal(loop(x0), want[1])


WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.pytorch import noise as PN
from skrample_amd.sampling import models as PM, structured as PT
from skrample_amd.sharding import BatchShard
name, global_batch, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
shard = BatchShard.from_env(global_batch // int(os.environ.get("WORLD_SIZE", "1")))   # RANK / WORLD_SIZE / LOCAL_RANK as torchrun sets them
dev = torch.device("cuda", 0)                       # (one card on this box: every rank uses it)
mk = {{
    "cfg4": lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()),
    "cfg5": lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps()),
    "cfg2": lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())),
}}[name]
g = torch.Generator().manual_seed(1234)
shape = (global_batch, 4, 32, 32)
x = torch.randn(shape, generator=g).bfloat16()
net = [torch.randn(shape, generator=g).bfloat16() for _ in range(3)]
lo, hi = shard.first_sample, shard.first_sample + shard.batch
seeds = shard.seeds(42)
w = mk()
w.set_timesteps(4)
cur = x[lo:hi].to(dev)
for k, t in enumerate(w.timesteps):
    o = (net[k % 3][lo:hi].to(dev) * 0.25 + cur * 0.5).bfloat16()
    cur = w.step(o, t, cur, generator=seeds, return_dict=False)[0]
torch.save({{"rank": shard.rank, "lo": lo, "hi": hi, "x": cur.cpu()}}, path)
"""


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg5"])
def test_two_process_shards_equal_the_single_process_run(name, dev, tmp_path):
    """SURVEY 8(e) dress rehearsal on one card: two FRESH child processes (plain subprocess children, never an exec of this
    process), each given RANK / WORLD_SIZE / LOCAL_RANK, take their `BatchShard.from_env` slice of a B = 8 batch through the HIP
    path and write it to a file; the concatenation equals the one-process run bit for bit.  (The scaling curve itself stays
    unmeasured: there is one GPU here.)"""
    from skrample_amd.sharding import BatchShard

    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    B = 8

    def launch(rank, world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        out = tmp_path / f"{name}_{world}_{rank}.pt"
        return subprocess.Popen([sys.executable, str(script), name, str(B), str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), out

    procs = [launch(r, 2) for r in range(2)]  # both at once: two processes on the card together, as two ranks of a node would be
    single, single_out = launch(0, 1)
    for p, _ in procs + [(single, single_out)]:
        log = p.communicate(timeout=600)[0].decode()
        assert p.returncode == 0, log
    whole = torch.load(single_out)
    parts = sorted((torch.load(o) for _, o in procs), key=lambda d: d["rank"])
    assert (whole["lo"], whole["hi"]) == (0, B)
    assert [(d["lo"], d["hi"]) for d in parts] == [(0, B // 2), (B // 2, B)]
    assert torch.equal(torch.cat([d["x"] for d in parts]), whole["x"])
    assert BatchShard(0, 2, B // 2).seeds(42) + BatchShard(1, 2, B // 2).seeds(42) == BatchShard(0, 1, B).seeds(42)


# ---- SPC's signed-power blend: a condition-aware bound instead of pinned seeds ------------------------
# blend(a, c) = spowf(wp * spowf(a, P) + wc * spowf(c, P), 1 / P)        reference structured.py:568-572, common.py:187-190
# Where the two powered terms cancel, the inner sum u carries the rounding of BOTH terms, |du| <= k * eps32 * (|t1| + |t2|), and
# the outer power turns it into |d blend| ~ (1/P) |u|^(1/P - 1) |du|: an absolute error that does not scale with the result.  A
# float32 evaluation -- the oracle's torch.pow as much as the kernel's exp2 / log2 -- cannot do better, so the parity bar for this
# one non-linear op is stated elementwise:
#     |hip - exact| <= 1e-5 * max|exact| + K * eps32 * (|wp| |a|^P + |wc| |c|^P)^(1/P)
# (for P < 1 the bracket majorises (1/P) |u|^(1/P-1) (|t1|+|t2|); for P > 1 the cancelled region is where the reference itself
# loses its digits and the bound is checked against the float64 value).
def _spow(v, f):
    return v.abs().pow(f) * torch.where(v < 0, -1.0, 1.0).to(v.dtype)


@pytest.mark.parametrize("power", [0.5, 2.0, 3.0, 1.0 / 3.0, 0.75])
def test_power_blend_error_model(power, dev):
    from skrample_amd.sampling import lazy

    eps32, K = 2.0**-23, 64.0
    worst = 0.0
    for seed in range(24):
        g = torch.Generator().manual_seed(7000 + seed)
        n = 4096
        a = torch.randn(n, generator=g)
        c = torch.randn(n, generator=g)
        wp, wc = (0.35, 0.65) if seed % 3 else (1.7, -0.7)  # (a negative weight: cancellation between same-signed operands)
        if seed % 2:  # constructed cancellation: the two powered terms nearly annihilate in a quarter of the elements
            t = (abs(wp) * a[: n // 4].abs().pow(power) / abs(wc)).pow(1.0 / power) * (1 + 1e-4 * torch.randn(n // 4, generator=g))
            c[: n // 4] = t * torch.where((a[: n // 4] < 0) ^ (wp * wc > 0), 1.0, -1.0)
        got = lazy.power_blend(a.to(dev), c.to(dev), wp, wc, power, torch.float32).cpu().double()
        a64, c64 = a.double(), c.double()
        exact = _spow(wp * _spow(a64, power) + wc * _spow(c64, power), 1.0 / power)
        scale = (abs(wp) * a64.abs().pow(power) + abs(wc) * c64.abs().pow(power)).pow(1.0 / power)
        bound = 1e-5 * exact.abs().max() + K * eps32 * scale
        if power > 1:  # sqrt-like outer power: |d blend| = |du| / (P |u|^(1 - 1/P)) is unbounded at u = 0; bound it through u
            u = wp * _spow(a64, power) + wc * _spow(c64, power)
            du = K * eps32 * (abs(wp) * a64.abs().pow(power) + abs(wc) * c64.abs().pow(power))
            bound = 1e-5 * exact.abs().max() + torch.maximum(_spow(u.abs() + du, 1.0 / power) - _spow((u.abs() - du).clamp_min(0), 1.0 / power), K * eps32 * scale)
        err = (got - exact).abs()
        assert (err <= bound).all(), (power, seed, (err / bound).max().item())
        worst = max(worst, (err / bound).max().item())
        # the oracle's own float32 evaluation sits under the same bound
        ref32 = _spow(wp * _spow(a, power) + wc * _spow(c, power), 1.0 / power).double()
        assert ((ref32 - exact).abs() <= bound).all()
    assert worst < 1.0


SPC_POWER = {
    "spc_power_half_dpm": (lambda: OA.make("spc", power=0.5, predictor=OA.make("dpm", 2, eta=0.5), corrector=OA.make("adams", 2)), lambda: PT.SPC(power=0.5, predictor=PT.DPM(order=2, stochasticity=0.5), corrector=PT.Adams(order=2))),
    "spc_power2": (lambda: OA.make("spc", power=2), lambda: PT.SPC(power=2)),
    "spc_power3_bias": (lambda: OA.make("spc", power=3, bias=0.2), lambda: PT.SPC(power=3, bias=0.2)),
}


@pytest.mark.parametrize("name", sorted(SPC_POWER))
def test_spc_power_seed_sweep_vs_oracle(name, dev):
    """The seed sweep the round-2 suite stopped drawing (a process-dependent `hash()` seed once produced 1.14e-5 against the 1e-5
    bar for `spc_power_half_dpm/scaled/data step 7`; that input is not recoverable, so 40 seeds x 4 schedule/model pairs x 9
    steps are drawn here instead).  Yardstick: the reference's arithmetic itself, evaluated in float64 -- the engine must be
    within the float32 bar of it, widened only by what the reference's OWN float32 evaluation loses to the blend's cancellation
    on the same input (its distance from its float64 self)."""
    from cases import MODELS, SCHEDULES, oracle_schedule
    from test_step_gpu import Injected

    mk_o, mk_p = SPC_POWER[name]
    steps, shape = 9, (3, 4, 24, 20)
    over_plain_bar = 0
    for seed in range(40):
        for sname, mname in (("karras_scaled", "eps"), ("linear", "flow"), ("zsnr", "v"), ("scaled", "data")):
            g = torch.Generator().manual_seed(100003 * seed + len(sname) * 31 + len(mname))
            w = PD.SkrampleWrapperScheduler(mk_p(), SCHEDULES[sname][1](), MODELS[mname][1])
            o32 = OW.StepDriver(mk_o(), oracle_schedule(sname, steps), MODELS[mname][0])
            o64 = OW.StepDriver(mk_o(), oracle_schedule(sname, steps), MODELS[mname][0], compute=torch.float64)
            for d in (w, o32, o64):
                d.set_timesteps(steps)
            noises = [torch.randn(shape, generator=g) for _ in range(steps)]
            w._noise_generator = Injected(noises, dev)
            x = torch.randn(shape, generator=g)
            for i, t in enumerate(w.timesteps):
                out = torch.randn(shape, generator=g)
                got = w.step(out.to(dev), t, x.to(dev), return_dict=False)[0].cpu().double()
                r32 = o32.step(out, t, x, noise=noises[i])[0]
                r64 = o64.step(out.double(), t, x.double(), noise=noises[i].double())[0]
                top = r64.abs().max().item()
                own = (r32.double() - r64).abs().max().item()  # what float32 costs the reference on this input
                err = (got - r64).abs().max().item()
                assert err <= 1e-5 * top + 4.0 * own, (name, seed, sname, mname, i, err / top, own / top)
                over_plain_bar += (got - r32.double()).abs().max().item() > 1e-5 * top
                x = r32
    # informational guard: excursions over the plain 1e-5 bar stay rare (they are the cancellation cases the model explains)
    assert over_plain_bar <= 8, over_plain_bar
