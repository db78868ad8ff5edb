#!/usr/bin/env python3
"""Benchmark of the fused sampler step.  Default workload = BASELINE.json's headline: DPM-2 SDE (eps-prediction, Karras(Scaled)
sigmas) over B x 4 x 128 x 128 bf16 latents, north-star shape B = 256 per GPU.  `--config` selects any other BASELINE config:

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --config cfg3c          # headline | cfg2 | cfg3 | cfg3c | cfg4 | cfg5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--config cfg4]

A "step" is one solver step over one per-GPU batch resident in HBM: the launches one step of the scheduler wrapper makes, issued
through the C ABI (skr_step_launch; for the configs that name a noise generator also that generator's launches).  The launch
plans are not hand-written here: the real scheduler wrapper (skrample_amd.diffusers.*WrapperScheduler) is run once over a
schedule with launch tracing on, and the launches it emitted for the steady-state steps are replayed on rotating buffer sets
(footprint well beyond the 256 MB Infinity Cache; every operand of a step has its own buffer in every set).  Multi-GPU = batch
shards with no collective (weak scaling: the per-GPU batch is fixed; configs 4 and 5 are DEFINED as 8-GPU shards -- B = 2048 / 8
and 512 / 8 per GPU -- so a run at N = 8 is the BASELINE configuration itself); noise seeds are indexed by global sample id.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""

from __future__ import annotations

import argparse
import ctypes
import dataclasses
import json
import math
import os
import sys
import time
from typing import Callable

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md:36)
HBM_ACHIEVABLE_GBS = 6300.0  # what that guide measures as achievable on this part (float4 copy, "8 TB/s peak (spec); ~6.3 TB/s achievable")


# ---------------------------------------------------------------------------------------------------------------------------
# workloads: the BASELINE.json configs (configs[0] is the CPU plumbing case: a parity test, not a bench line)
# ---------------------------------------------------------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class Workload:
    name: str
    title: str  # config.workload text ({batch} = per-GPU batch)
    metric: str
    batch: int  # samples per GPU
    unit: tuple[int, int, int]  # per-sample latent shape
    bytes_per_elem: int  # algorithmic bytes per element per solver step (step kernels only)
    calls: int  # wrapper.step() calls per solver step (Runge-Kutta: one per stage)
    schedule_steps: int
    steady: tuple[int, ...]  # solver-step indices of the traced schedule that are cycled
    make: Callable  # () -> scheduler wrapper
    kernel: str  # roofline.kernel text
    pmc_kernels: tuple[tuple[str, int], ...]  # (kernel-name needle, launches per solver step) for the live traffic passes
    mix_ceiling: dict | None = None
    generator_bytes_note: str | None = None
    oracle: Callable | None = None  # (steps) -> (driver, stepper) for cpu_baseline
    cpu_sample: int = 64


def _wl() -> dict[str, Workload]:
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.pytorch import noise as PN
    from skrample_amd.sampling import models as PM
    from skrample_amd.sampling import structured as PT

    # Default-constructed wrappers (alias_history="auto"): a run's first call snapshots its two tensors, the second call sees the
    # synthetic "network" hand over a fresh tensor and aliases from then on -- the traced steady-state steps (5-14) are the aliased ones
    dpm2 = lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()))  # noqa: E731
    steady = tuple(range(5, 15))
    return {
        w.name: w
        for w in (
            Workload(
                "headline", "DPM order-2 SDE (eta=1) + Karras(Scaled) sigmas, eps-pred, B={batch}x4x128x128 bf16 latents per GPU "
                "(BASELINE north-star shape; cfg2 at 4x batch), in-kernel Philox noise, one fused launch per step",
                "sampler steps/sec (fused DPM-2 SDE step, eps-pred, Karras sigmas, Bx4x128x128 bf16) + achieved HBM GB/s",
                256, (4, 128, 128), 10, 1, 20, steady, dpm2,
                "skr::step_kernel_k1<bf16_t, K=4, NOISE=true> (one-trip, paced loads, XCD chunk map)",
                (("step_kernel_k1<skr::bf16_t, 4, true", 1),),
                {"us_per_launch": 25.76, "frac": 0.814, "source": "profiles/r03_harness_lib_vs_ceilings.txt (kmix<R4,W1>, committed harness run)"},
                oracle=lambda: _step_oracle("dpm", 2, 1.0, "karras_scaled", "eps", "random"),
            ),
            Workload(
                "cfg2", "BASELINE config 2: DPM order-2 SDE (eta=1) + Karras(Scaled) sigmas, eps-pred, B={batch}x4x128x128 bf16 on 1 MI355X, "
                "in-kernel Philox noise, one fused launch per step (launch-bound at this size; the graph-captured loop is `graph_loop_cfg2` of the headline line)",
                "sampler steps/sec (fused DPM-2 SDE step, eps-pred, Karras sigmas, 64x4x128x128 bf16) + achieved HBM GB/s",
                64, (4, 128, 128), 10, 1, 20, steady, dpm2,
                "skr::step_kernel_k1<bf16_t, K=4, NOISE=true>",
                (("step_kernel_k1<skr::bf16_t, 4, true", 1),),
                {"us_per_launch": 8.34, "frac": 0.628, "source": "profiles/r03_harness_placement_phase_b64.txt (no-arithmetic 4r+1w kernel at B=64, launch to launch)"},
                oracle=lambda: _step_oracle("dpm", 2, 1.0, "karras_scaled", "eps", "random"),
            ),
            Workload(
                "cfg3", "BASELINE config 3 with white noise: UniPC order-3 SDE (eta=1), flow-pred, Linear schedule, B={batch}x16x128x128 bf16, "
                "in-kernel Philox noise (two draws per element), one two-output launch per step (fp32 corrected state + bf16 result)",
                "sampler steps/sec (fused UniPC-3 SDE step, flow-pred, Linear schedule, Bx16x128x128 bf16, Philox noise) + achieved HBM GB/s",
                256, (16, 128, 128), 26, 1, 20, steady,
                lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel()),
                "skr::step_kernel_k2<bf16_t, 8 + 1 operands, NOISE=true> (two outputs: fp32 state + bf16 result)",
                (("step_kernel_k2<skr::bf16_t, 8, 1, true", 1),),
                oracle=lambda: _step_oracle("unipc", 3, 1.0, "linear", "flow", "random"), cpu_sample=16,
            ),
            Workload(
                "cfg3c", "BASELINE config 3: UniPC order-3 SDE (eta=1) + Colored noise, flow-pred, Linear schedule, B={batch}x16x128x128 bf16 on 1 MI355X: "
                "per step one Colored draw (3 launches: plane FFTs, channel axis + radial weights, inverse planes) and one two-output step launch",
                "sampler steps/sec (UniPC-3 SDE step + Colored noise, flow-pred, Linear schedule, 256x16x128x128 bf16) + achieved HBM GB/s",
                256, (16, 128, 128), 30, 1, 20, steady,
                lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps()),
                "skr::step_kernel_k2<bf16_t, 10 + 1 operands> (two outputs); generator: colored_plane<0> + colored_outer_axis_regs<16> + colored_plane<1>",
                (("step_kernel_k2<skr::bf16_t, 10, 1, false", 1), ("colored_plane", 2), ("colored_outer_axis", 1)),
                generator_bytes_note="Colored generator, unavoidable HBM traffic per draw: half spectrum (complex64) written, read + written by the channel-axis pass, read by the inverse, + bf16 result = 18.2 B/element",
                oracle=lambda: _step_oracle("unipc", 3, 1.0, "linear", "flow", "colored"), cpu_sample=8,
            ),
            Workload(
                "cfg4", "BASELINE config 4: Adams/IPNDM order-4 ODE, v-pred, ZSNR schedule, B=2048x4x128x128 bf16 batch-sharded across 8 MI355X "
                "= {batch} samples per GPU (the shard a rank owns; no collective), one fused launch per step",
                "sampler steps/sec (fused Adams-4 step, v-pred, ZSNR, 2048x4x128x128 bf16 sharded by sample over 8 GPUs: 256 per GPU) + achieved HBM GB/s",
                256, (4, 128, 128), 18, 1, 20, steady,
                lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()),
                "skr::step_kernel_k1<bf16_t, K=8> (x, out and three verbatim (x, out) history pairs)",
                (("step_kernel_k1<skr::bf16_t, 8, false", 1),),
                oracle=lambda: _step_oracle("adams", 4, 0.0, "zsnr", "v", "random"),
            ),
            Workload(
                "cfg5", "BASELINE config 5: RKUltra order-6 (Cash-Karp, 6 stages) SDE (eta=1) + Pyramid noise, eps-pred, Scaled schedule, B=512x4x256x256 bf16 "
                "batch-sharded across 8 MI355X = {batch} samples per GPU; a step = 6 stage launches (rounded derivative conversion + next stage input) "
                "+ one Pyramid draw (2 launches)",
                "sampler steps/sec (RKUltra-6 SDE step = 6 fused stage launches + Pyramid noise, 512x4x256x256 bf16 sharded by sample over 8 GPUs: 64 per GPU) + achieved HBM GB/s",
                64, (4, 256, 256), 100, 6, 6, (1, 2, 3, 4),
                lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps()),
                "skr::step_kernel_rk1<bf16_t, K=2..7> x 6 stages (derivative + next stage input per launch); generator: pyramid_pass1 + normalise_pass2",
                (("step_kernel_rk1", 6), ("pyramid_pass1", 1), ("normalise_pass2", 1)),
                generator_bytes_note="Pyramid generator: fp32 scratch written and re-read by the normalising pass + bf16 result = 10 B/element per draw",
                oracle=lambda: _rk_oracle(), cpu_sample=8,
            ),
        )
    }


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed solver steps (default 400; 100 for cfg3 / cfg3c / cfg5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: a tenth of --steps)")
    ap.add_argument("--config", default="headline", choices=["headline", "cfg2", "cfg3", "cfg3c", "cfg4", "cfg5"], help="BASELINE.json workload (default: the headline DPM-2 launch at B=256)")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the config's own)")
    ap.add_argument("--sets", type=int, default=6, help="rotating buffer sets (>= 4; raised until the footprint passes 1 GB)")
    ap.add_argument("--precondition", type=int, default=None, help="untimed conditioning steps before the warm-up (default ~80 ms worth; see the comment at its use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 FETCH_SIZE / WRITE_SIZE passes (roofline.traffic falls back to the committed summary)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="diagnostic: kernel-selection switches (skr_set_tuning) applied before anything is launched")
    ap.add_argument("--drift", type=int, default=0, help="diagnostic: run this many steps back to back and print the average step time of every block of 50 (then exit)")
    ap.add_argument("--no-extras", action="store_true", help="no wrapper-rate, graph-loop and memcpy keys (used by the counter passes)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------------
# capture: run the real wrapper once, lift the launches of the steady-state steps
# ---------------------------------------------------------------------------------------------------------------------------
class _Seed:
    "minimal stand-in for torch.Generator as a seed carrier (initial_seed only)"

    def __init__(self, s: int):
        self.s = s

    def initial_seed(self) -> int:
        return self.s


@dataclasses.dataclass
class Launch:
    plan: object
    inputs: list[int]  # slot ids
    out0: int | None
    out1: int | None
    seeds: object  # device tensor or None
    numel: int


@dataclasses.dataclass
class StepRecord:
    launches: list[Launch]
    slots: dict[int, tuple[int, torch.dtype, str]]  # slot id -> (numel, dtype, placement: "net" | "engine" | "noise")
    drawn: int | None  # slot written by this step's generator call
    noise_step: object  # the Step the generator was asked for


def capture(wl: Workload, batch: int, dev: torch.device, shard) -> tuple[dict[int, StepRecord], object]:
    """Run the real wrapper over one schedule with tracing on.  Per steady-state solver step: its launches (the filled plans) with
    every operand named by a slot; which slot the network produced (placed by torch's allocator in the replay, as in a run),
    which the engine allocated, and which one this step's noise generator wrote."""
    from skrample_amd import _hip
    from skrample_amd.pytorch import noise as PN

    w = wl.make()
    w.set_timesteps(wl.schedule_steps)
    g = torch.Generator(device=dev).manual_seed(shard.input_seed())
    shape = (batch, *wl.unit)
    gens = [_Seed(s) for s in shard.seeds()]  # per-sample seeds by GLOBAL sample index (skrample_amd/sharding.py)
    x = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
    drawn_log: list[tuple[torch.Tensor, object]] = []
    original = PN.BatchTensorNoise.generate_lazy

    def recording(self, step):
        item = original(self, step)
        if isinstance(item, torch.Tensor):
            drawn_log.append((item, step))
        return item

    keep: list = [x]  # every tensor of the traced run stays alive: an address names one tensor
    net: set[int] = set()
    per_step: dict[int, list] = {}
    draws: dict[int, list] = {}
    PN.BatchTensorNoise.generate_lazy = recording
    _hip.trace = []
    try:
        for c, t in enumerate(w.timesteps):
            out = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
            net.add(out.data_ptr())
            _hip.trace.clear()
            drawn_log.clear()
            nxt = w.step(out, t, x, generator=gens, return_dict=False)[0]
            s = c // wl.calls
            per_step.setdefault(s, []).extend(_hip.trace)
            draws.setdefault(s, []).extend(drawn_log)
            keep.extend([out, nxt, list(_hip.trace), list(drawn_log)])
            x = nxt
    finally:
        _hip.trace = None
        PN.BatchTensorNoise.generate_lazy = original
    torch.cuda.synchronize(dev)
    records: dict[int, StepRecord] = {}
    for s in wl.steady:
        entries = per_step[s]
        assert len(entries) == wl.calls, f"step {s}: {len(entries)} launches for {wl.calls} wrapper calls (a solver step must be one fused launch per call)"
        ids: dict[int, int] = {}
        slots: dict[int, tuple[int, torch.dtype, str]] = {}
        noise_ptrs = {t.data_ptr() for t, _ in draws.get(s, [])}

        def slot(t: torch.Tensor | None) -> int | None:
            if t is None:
                return None
            p = t.data_ptr()
            if p not in ids:
                ids[p] = len(ids)
                slots[ids[p]] = (t.numel(), t.dtype, "noise" if p in noise_ptrs else "net" if p in net else "engine")
            return ids[p]

        launches = [Launch(plan, [slot(t) for t in inputs], slot(o0), slot(o1), seeds, numel) for plan, inputs, o0, o1, seeds, numel in entries]
        assert all(l.numel == batch * math.prod(wl.unit) for l in launches)
        assert len(draws.get(s, [])) <= 1
        drawn = ids[draws[s][0][0].data_ptr()] if draws.get(s) else None
        records[s] = StepRecord(launches, slots, drawn, draws[s][0][1] if draws.get(s) else None)
    return records, w


class _HipEvent:
    """A raw HIP event created with hipEventDisableSystemFence.  torch.cuda.Event records carry a system-scope release (cache
    write-back + invalidate): harmless at the ends of the timed region, but as a marker BETWEEN two launches it opens a bubble of
    several microseconds (24 us under rocprofv3) that would be billed to the kernels.  Same API subset as torch.cuda.Event."""

    _hip = None

    @classmethod
    def runtime(cls):
        if cls._hip is None:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
            hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            hip.hipEventQuery.argtypes = [ctypes.c_void_p]
            hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
            hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
            cls._hip = hip
        return cls._hip

    def __init__(self, stream: int):
        self.stream = stream
        self.handle = ctypes.c_void_p()
        if self.runtime().hipEventCreateWithFlags(ctypes.byref(self.handle), 0x20000000) != 0:  # hipEventDisableSystemFence
            raise RuntimeError("hipEventCreateWithFlags failed")

    def record(self) -> None:
        if self._hip.hipEventRecord(self.handle, self.stream) != 0:
            raise RuntimeError("hipEventRecord failed")

    def query(self) -> bool:
        return self._hip.hipEventQuery(self.handle) == 0

    def elapsed_time(self, other: "_HipEvent") -> float:
        ms = ctypes.c_float()
        self._hip.hipEventSynchronize(other.handle)
        if self._hip.hipEventElapsedTime(ctypes.byref(ms), self.handle, other.handle) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        return ms.value


# ---------------------------------------------------------------------------------------------------------------------------
# cpu_baseline: the oracle's reference-order port of the same workload, on a bounded sample
# ---------------------------------------------------------------------------------------------------------------------------
def _oracle_modules():
    if os.path.join(ROOT, "oracle") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from skr_oracle import noise as ON
    from skr_oracle import rk as OK
    from skr_oracle import samplers as OA
    from skr_oracle import schedules as OS
    from skr_oracle import wrapper as OW

    return ON, OK, OA, OS, OW


def _step_oracle(kind: str, order: int, eta: float, schedule: str, pred: str, noise: str):
    "(driver, one_call) of a multistep config: the oracle's StepDriver in the reference's op order, deep copies included"
    ON, OK, OA, OS, OW = _oracle_modules()

    def build(steps: int, seeds: list[int]):
        sched = {"karras_scaled": lambda: OS.karras(OS.scaled(), steps=steps), "linear": OS.linear, "zsnr": OS.zsnr}[schedule]()
        drv = OW.StepDriver(OA.make(kind, order, eta=eta), sched, pred, noise_kind=noise, mimic_copies=True)
        return drv, (lambda out, t, x: drv.step(out, t, x, seeds=seeds)[0])

    what = f"oracle StepDriver ({kind}-{order}, eta={eta:g}, {schedule}, {pred}-pred, {noise} noise; reference op order, fp32 compute, per-sample generators + stack, deep copies)"
    return build, what


def _rk_oracle():
    "(driver, one_call) of BASELINE config 5: the oracle's inside-out Runge-Kutta driver + per-sample Pyramid noise"
    ON, OK, OA, OS, OW = _oracle_modules()

    def build(steps: int, seeds: list[int]):
        drv = OW.RKDriver(OK.pick_tableau(6), OS.scaled(), "eps", "data", 1.0)
        gens = [ON.torch_draws(torch.Generator().manual_seed(int(s))) for s in seeds]
        unit = [None]

        def noise_fn(step=None):  # diffusers.py:312-346: one generator per batch item, stacked
            return torch.stack([ON.pyramid_noise(unit[0], randn, rand1) for randn, rand1 in gens])

        def call(out, t, x):
            unit[0] = tuple(x.shape[1:])
            return drv.step(out, t, x, noise_fn=noise_fn)

        return drv, call

    return build, "oracle RKDriver (Cash-Karp 6 stages, eta=1, Scaled, eps-pred, derivative space = data, per-sample Pyramid generators + stack; reference op order, fp32 compute)"


def _cpu_port_rate(wl: Workload, seconds: float, threads: int) -> tuple[float, int, int]:
    "per-GPU-batch-equivalent solver steps/s of the oracle's reference-order port with `threads` torch threads"
    torch.set_num_threads(threads)
    sub = min(wl.cpu_sample, wl.batch)  # the port's cost is linear in the batch (per-sample generators, elementwise passes)
    build, _ = wl.oracle()
    g = torch.Generator().manual_seed(1234)
    seeds = [42 + i for i in range(sub)]
    times: list[float] = []
    t_start = time.perf_counter()
    schedules = 0
    shape = (sub, *wl.unit)
    while time.perf_counter() - t_start < seconds:  # whole schedules until ~`seconds` of CPU work
        drv, call = build(wl.schedule_steps, seeds)
        drv.set_timesteps(wl.schedule_steps)
        x = torch.randn(shape, generator=g).bfloat16()
        acc = 0.0
        for c, t in enumerate(drv.timesteps):
            out = torch.randn(shape, generator=g).bfloat16()
            t0 = time.perf_counter()
            x = call(out, t, x)
            acc += time.perf_counter() - t0
            if (c + 1) % wl.calls == 0:
                if c // wl.calls in wl.steady:
                    times.append(acc)
                acc = 0.0
            if time.perf_counter() - t_start > 2.5 * seconds and times:
                break
        schedules += 1
    per_step_full = (sum(times) / len(times)) * (wl.batch / sub)
    return 1.0 / per_step_full, len(times), schedules


def cpu_baseline(wl: Workload, seconds: float, world: int) -> dict:
    """The oracle (reference-order torch CPU port, incl. the reference's deep copies and per-sample randn + stack) timed on this box's
    host cores on a bounded sample: a slice of the batch, scaled to the per-GPU batch.  Timed twice -- with torch's default thread count
    and with 16 threads (elementwise passes over a few MB do not scale to a whole socket) -- and the faster of the two is reported."""
    default_threads = torch.get_num_threads()
    trials = {}
    # (torch.distributed.run exports OMP_NUM_THREADS=1: the 16-thread trial is taken from the machine's core count, not from torch's default)
    for threads in sorted({default_threads, min(16, max(default_threads, os.cpu_count() or 1))}):
        trials[threads] = _cpu_port_rate(wl, seconds / 2, threads)
    torch.set_num_threads(default_threads)
    best = max(trials, key=lambda k: trials[k][0])
    rate, nsteps, schedules = trials[best]
    others = ", ".join(f"{k} threads: {v[0]:.3g} steps/s" for k, v in trials.items())
    sub = min(wl.cpu_sample, wl.batch)
    return {
        "value": rate,
        "unit": "steps/s",
        "cores": best,
        "kind": "port",
        "sample": f"{wl.oracle()[1]} on {sub} of {wl.batch} samples, {nsteps} steady-state steps over {schedules} runs of a {wl.schedule_steps}-step schedule, "
        f"time scaled x{wl.batch / sub:g} to B={wl.batch}; best of ({others})" + (f"; timed on rank 0 while the other {world - 1} ranks wait at a barrier" if world > 1 else ""),
    }


# ---------------------------------------------------------------------------------------------------------------------------
# extras of the headline line (informational keys)
# ---------------------------------------------------------------------------------------------------------------------------
def graph_loop_rate(dev: torch.device) -> dict | None:
    """SURVEY 8(f) rank 1, reported beside the headline: BASELINE config 2 itself (B=64x4x128x128, DPM-2 SDE, Karras, 20 steps)
    as ONE HIP graph of the whole sampler loop with device-resident step scalars (skr_step_launch_indexed) -- the launch-bound
    regime where the eager wrapper is host-limited.  Sampler only: the "network" hands back pre-generated tensors."""
    try:
        import skrample_amd.diffusers as PD
        import skrample_amd.scheduling as PS
        from skrample_amd.graphs import capture_sampling_loop
        from skrample_amd.sampling import structured as PT

        batch, steps = 64, 20
        shape = (batch, 4, 128, 128)
        g = torch.Generator(device=dev).manual_seed(7)
        x0 = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
        outs = [torch.randn(shape, device=dev, generator=g).to(torch.bfloat16) for _ in range(4)]
        calls = [0]

        def net(x, t):  # distinct buffers in turn (the wrapper guards its aliased history against reused output buffers)
            calls[0] += 1
            return outs[calls[0] % len(outs)]

        seeds = list(range(42, 42 + batch))
        # (alias_history=True: `net` rotates distinct buffers, so no snapshot kernels are recorded into the loop; the default "auto"
        #  would snapshot the first call of every run -- two copy kernels per replay)
        mk = lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), alias_history=True)  # noqa: E731
        w = mk()

        def eager():
            w.set_timesteps(steps)
            x = x0
            for t in w.timesteps.tolist():
                x = w.step(net(x, t), t, x, generator=seeds, return_dict=False)[0]
            return x

        for _ in range(3):
            eager()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            eager()
        torch.cuda.synchronize(dev)
        eager_s = (time.perf_counter() - t0) / 10
        loop = capture_sampling_loop(mk(), net, x0, steps, seeds=seeds, indexed=True)
        for _ in range(5):
            loop.graph.replay()
        torch.cuda.synchronize(dev)
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            loop.graph.replay()
        torch.cuda.synchronize(dev)
        graph_s = (time.perf_counter() - t0) / reps
        return {
            "workload": f"BASELINE config 2: DPM-2 SDE + Karras, eps-pred, B={batch}x4x128x128 bf16, {steps}-step loop, sampler only",
            "graph_steps_per_s": steps / graph_s,
            "graph_us_per_step": graph_s / steps * 1e6,
            "eager_steps_per_s": steps / eager_s,
            "eager_us_per_step": eager_s / steps * 1e6,
            "mode": "one HIP graph per loop, step scalars read from a device-resident table (skr_step_launch_indexed)",
        }
    except Exception as exc:  # informational key: never take the headline down
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


def wrapper_rate(wl: Workload, batch: int, dev: torch.device, shard, nbuf: int = 6) -> float | None:
    "solver steps/s through the eager Python scheduler wrapper (host overhead included; first call of a run excluded), for information"
    try:
        w = wl.make()
        shape = (batch, *wl.unit)
        g = torch.Generator(device=dev).manual_seed(5)
        xs = [torch.randn(shape, device=dev, generator=g).to(torch.bfloat16) for _ in range(nbuf)]
        outs = [torch.randn(shape, device=dev, generator=g).to(torch.bfloat16) for _ in range(nbuf)]
        seeds = [_Seed(sd) for sd in shard.seeds()]
        rate = None
        for _ in range(3):
            w.set_timesteps(wl.schedule_steps if wl.calls > 1 else max(wl.schedule_steps, 100))  # (a longer window: start / stop of the region spread over more steps)
            ts = w.timesteps.tolist()
            ts = ts[: len(ts) // wl.calls * wl.calls]
            x = w.step(outs[0], ts[0], xs[0], generator=seeds, return_dict=False)[0]  # first call of a run also builds the per-sample generators
            torch.cuda.synchronize(dev)
            tw = time.perf_counter()
            for i, t in enumerate(ts[1:], start=1):
                # multistep: fresh sample + output buffers per call; Runge-Kutta: the stage input the wrapper returned
                x = w.step(outs[i % nbuf], t, x if wl.calls > 1 else xs[i % nbuf], generator=seeds, return_dict=False)[0]
            torch.cuda.synchronize(dev)
            rate = (len(ts) - 1) / wl.calls / (time.perf_counter() - tw)
        return rate
    except Exception as exc:  # informational
        print(f"[bench] wrapper rate failed: {type(exc).__name__}: {exc}", file=sys.stderr)
        return None


# ---------------------------------------------------------------------------------------------------------------------------
# HBM traffic from the PMC counters
# ---------------------------------------------------------------------------------------------------------------------------
def load_traffic(wl: Workload) -> tuple[float | None, str]:
    """HBM bytes per solver step from the committed PMC passes of this same command (profiles/r05_pmc_traffic_<config>.json, else the
    headline's earlier rounds): the fallback when the live passes below cannot run"""
    names = [f"r05_pmc_traffic_{wl.name}.json", f"r04_pmc_traffic_{wl.name}.json"] + (["r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"] if wl.name == "headline" else [])
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.isfile(path):
            try:
                d = json.load(open(path))
                return float(d.get("hbm_bytes_per_step", d.get("hbm_bytes_per_launch"))), f"profiles/{name} (committed summary of the same passes)"
            except Exception:
                continue
    return None, "no committed summary"


def measure_traffic(wl: Workload, batch: int, steps: int = 40) -> tuple[float | None, str, dict]:
    """HBM bytes per solver step, measured now: this command is re-run twice as a child process under
    `rocprofv3 --pmc <counter> --kernel-trace` (FETCH_SIZE and WRITE_SIZE in separate passes, as the guide's HBM section
    prescribes).  Per kernel of the workload (name needle, launches per step) the counter rows of the LAST `steps x launches`
    dispatches -- the child's timed region, steady-state steps only -- are summed, and the guide's gfx950 corrections are applied
    (values are KiB; FETCH_SIZE counts the 128-B requests of wide coalesced streams as 64 B -> x2; WRITE_SIZE is exact).
    Returns (bytes per step, provenance, per-kernel breakdown); (None, reason, {}) when the profiler is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler", {}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", {}
    tmp = tempfile.mkdtemp(prefix="skr_pmc_", dir="/tmp")
    per_kernel: dict[str, dict[str, float]] = {}
    try:
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, name)
            cmd = [exe, "--pmc", name, "--kernel-trace", "-d", out_dir, "-o", "prof", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--config", wl.name, "--steps", str(steps), "--warmup", "2", "--precondition", "0",
                   "--batch", str(batch), "--no-cpu-baseline", "--no-traffic", "--no-extras"]  # fmt: skip
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
            env["TMPDIR"] = "/tmp"
            proc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=180)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if proc.returncode != 0 or not files:
                return None, f"{name} pass failed (rc {proc.returncode}): {proc.stderr.decode(errors='replace')[-160:]}", {}
            rows = [r for f in files for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            for needle, per_step in wl.pmc_kernels:
                vals = [float(r["Counter_Value"]) for r in rows if needle in r["Kernel_Name"]]
                want = steps * per_step
                if len(vals) < want:
                    return None, f"{name} pass: only {len(vals)} launches of {needle!r} found, {want} expected", {}
                per_kernel.setdefault(needle, {})[name] = sum(vals[-want:]) / steps  # KiB per solver step
    except subprocess.TimeoutExpired:
        return None, "counter pass timed out", {}
    except Exception as exc:
        return None, f"{type(exc).__name__}: {exc}"[:200], {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    breakdown = {k: 2.0 * v["FETCH_SIZE"] * 1024.0 + v["WRITE_SIZE"] * 1024.0 for k, v in per_kernel.items()}
    return sum(breakdown.values()), (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in child processes "
                                     f"(last {steps} solver steps of each; KiB, FETCH x2 on gfx950)"), breakdown


# ---------------------------------------------------------------------------------------------------------------------------
def main() -> None:
    args = parse()
    from skrample_amd.sharding import BatchShard, TimedRegion, aggregate_rate, max_over_ranks, rank_spread

    wl = _wl()[args.config]
    heavy = wl.name in ("cfg3", "cfg3c", "cfg5")
    if args.steps is None:
        args.steps = 100 if heavy else 400
    if args.warmup is None:
        args.warmup = max(args.steps // 10, 1)
    batch = args.batch or wl.batch
    shard = BatchShard.from_env(batch)
    rank, local_rank, world = shard.rank, shard.local_rank, shard.world
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU path)")
    # rehearsal knobs for a one-GPU box (the multi-GPU run proper is one rank per GPU over RCCL): SKR_BENCH_DEVICE pins every
    # rank to one device index, SKR_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device
    backend = os.environ.get("SKR_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", int(os.environ.get("SKR_BENCH_DEVICE", local_rank)))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # under torchrun, also for a single rank
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its communicator comes up (first collective): keep stdout for the one
        # JSON line by pointing fd 1 at stderr while the communicator is created
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from skrample_amd import _hip

    lib = _hip.load()
    for item in args.tune:
        key, _, val = item.partition("=")
        if lib.skr_set_tuning(key.encode(), int(val)) != 0:
            raise SystemExit(f"--tune {item}: refused")
    numel = batch * math.prod(wl.unit)
    records, traced_wrapper = capture(wl, batch, dev, shard)
    generator = traced_wrapper._noise_generator if any(r.drawn is not None for r in records.values()) else None

    # rotating buffer sets.  Every operand slot of a step gets its own buffer in every set.  In a sampling loop the sample, the
    # history samples and the results are tensors the engine itself allocated (step results), so they come from its output
    # allocator (lazy.empty_output: 4 KiB-staggered placement); network outputs come straight from torch's allocator.
    from skrample_amd.sampling.lazy import empty_output

    set_bytes = max(sum(n * dt.itemsize for n, dt, _ in rec.slots.values()) for rec in records.values())
    nsets = max(args.sets, 4, -(-(1 << 30) // set_bytes))
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    order = [s for s in wl.steady]
    plan_calls = []  # one entry per (steady step, buffer set) in cycling order
    held = []
    bufsets: list[dict[tuple[int, int], torch.Tensor]] = []
    for k in range(nsets):
        bufs: dict[tuple[int, int], torch.Tensor] = {}
        for s in order:
            rec = records[s]
            written = {l.out0 for l in rec.launches} | {l.out1 for l in rec.launches}
            for sid, (n, dt, place) in rec.slots.items():
                key = (n, dt, place, sid)  # the steady steps share one buffer per slot id within a set (same role in every step)
                if key in bufs:
                    continue
                if place == "engine":
                    t = empty_output((n,), dt, dev)
                    if sid not in written:
                        t.copy_(torch.randn(n, device=dev, generator=g))
                else:
                    t = torch.randn(n, device=dev, generator=g).to(dt)
                bufs[key] = t
        bufsets.append(bufs)
    held.append(bufsets)
    for k in range(len(order) * nsets):
        s = order[k % len(order)]
        rec = records[s]
        bufs = bufsets[k % nsets]
        buf = lambda sid: bufs[(*rec.slots[sid], sid)]  # noqa: E731
        calls = []
        for l in rec.launches:
            ptrs = (ctypes.c_void_p * max(len(l.inputs), 1))(*[buf(i).data_ptr() for i in l.inputs])
            patch = [j for j, i in enumerate(l.inputs) if i == rec.drawn]
            calls.append((ctypes.byref(l.plan), ptrs, buf(l.out0).data_ptr() if l.out0 is not None else None, buf(l.out1).data_ptr() if l.out1 is not None else None,
                          l.seeds.data_ptr() if l.seeds is not None else None, patch))  # fmt: skip
        plan_calls.append((calls, rec.noise_step if rec.drawn is not None else None))

    stream = torch.cuda.current_stream(dev).cuda_stream
    launches_per_step = wl.calls
    last_noise = [None, None]

    marked = [0]  # solver steps issued when `mark` was recorded

    def issue(call, fresh) -> None:
        p, ptrs, o0, o1, sd, patch = call
        if fresh is not None:
            for j in patch:
                ptrs[j] = fresh
        status = lib.skr_step_launch(p, ptrs, o0, o1, sd, numel, stream)
        if status:
            _hip.check(status, "skr_step_launch")

    def draw_noise(noise_step):
        fresh = generator.generate_lazy(noise_step)  # the wrapper's own call (diffusers.py::get_step_noise): 2-3 launches into a fresh tensor
        last_noise[0], last_noise[1] = fresh, last_noise[0]  # (the previous draw stays alive while it may still be read)
        return fresh.data_ptr()

    def run(count: int, offset: int = 0, mark=None, draw: bool = True) -> None:
        """issue `count` solver steps: the generator's launches (when the config names one and `draw`), then the step launches.
        Runge-Kutta configs (several wrapper calls per step) are issued STAGE-MAJOR over groups of `nsets` steps, one step per
        buffer set: between two stages of one step the network runs in a real pipeline, so a stage must not find the tensors the
        previous stage of its step just wrote still in the Infinity Cache -- here nsets - 1 launches on other sets lie in between."""
        n = len(plan_calls)
        if launches_per_step == 1:
            for i in range(count):
                calls, noise_step = plan_calls[(offset + i) % n]
                issue(calls[0], draw_noise(noise_step) if noise_step is not None and draw else None)
                if i == 0 and mark is not None:
                    mark.record()  # behind the first timed step: the steady-state clock excludes the cold-queue start of the region
                    marked[0] = 1
            return
        done = 0
        while done < count:
            group = [plan_calls[(offset + done + k) % n] for k in range(min(nsets, count - done))]
            for j in range(launches_per_step):
                for calls, noise_step in group:
                    fresh = draw_noise(noise_step) if j == launches_per_step - 1 and noise_step is not None and draw else None  # drawn at the step's last stage, as the wrapper does
                    issue(calls[j], fresh)
            done += len(group)
            if mark is not None and marked[0] == 0:
                mark.record()
                marked[0] = done

    try:  # HIP events on the launch stream, without the system-scope fence of torch's events (see _HipEvent)
        e0, e1, e_first, e_warm, f0, f1 = (_HipEvent(stream) for _ in range(6))
    except Exception:
        e0, e1, e_first, e_warm, f0, f1 = (torch.cuda.Event(enable_timing=True) for _ in range(6))
    for ev in (e0, e1, e_first, e_warm, f0, f1):  # (torch creates its HIP event at the first record: do that outside the timed region)
        ev.record()
    if args.drift:  # how the step time moves over a long back-to-back run (clock / power management), 50 steps per reading
        marks = [_HipEvent(stream) for _ in range(args.drift // 50 + 1)]
        torch.cuda.synchronize(dev)
        marks[0].record()
        for b in range(1, len(marks)):
            run(50, offset=50 * (b - 1))
            marks[b].record()
        torch.cuda.synchronize(dev)
        per = [marks[b - 1].elapsed_time(marks[b]) * 1e3 / 50 for b in range(1, len(marks))]
        print("us per step, blocks of 50:", " ".join(f"{v:.2f}" for v in per))
        tail = per[len(per) // 2 :]
        print(f"second half of the run: {sum(tail) / len(tail):.3f} us per step  (--tune {args.tune})")
        return
    # conditioning (untimed, before the contract's W warm-up steps): ~26 ms of back-to-back launches so that clocks and the page
    # tables of all buffer sets are in their steady state however small W and K are.  `--drift 4000` shows why it has to be this
    # long: from a cold start the launch time rises to 26.3-26.4 us between launches ~100 and ~350 (power management settling),
    # comes back by launch ~600 and then stays at 25.7-25.85 us for as long as the run lasts; a K=20 window opened after 300
    # launches sat in that transient.  Round 4: on other boxes the settling takes ~1500 launches (26.8 us for the first 500, 26.2-26.3
    # to ~1000, 26.0-26.1 from there on), so the conditioning is ~80 ms of launches now (3000 at the headline size).  The count is
    # reported in the JSON line (config.precondition_steps).
    ideal_us = numel * wl.bytes_per_elem / (HBM_PEAK_GBS * 1e3)
    precondition = args.precondition if args.precondition is not None else max(20, min(3000, int(63000.0 / ideal_us)))
    run(precondition)
    torch.cuda.synchronize(dev)
    stream_query = None  # hipStreamQuery while polling: the runtime retires finished launches during the wait, not inside synchronize()
    if not os.environ.get("SKR_BENCH_NO_STREAM_POLL"):
        try:
            stream_query = _HipEvent.runtime().hipStreamQuery
            stream_query.argtypes = [ctypes.c_void_p]
        except Exception:
            stream_query = None
    run(args.warmup)
    e_warm.record()
    while not e_warm.query():  # poll, then synchronize: a blocking wait returns tens of us late, and the GPU would sit idle
        if stream_query is not None:  # (and start the timed region from a colder state) for that long
            stream_query(stream)
    torch.cuda.synchronize(dev)
    # The contract's window: barrier + synchronise on both sides, but each rank reads its clock right after ITS OWN synchronise and
    # only then joins the closing barrier (sharding.TimedRegion) -- the engine has no collective, so none is billed to the K steps.
    region = TimedRegion(dist, sync=lambda: torch.cuda.synchronize(dev))
    t0 = region.open()
    e0.record()
    marked[0] = 0
    run(args.steps, offset=args.warmup, mark=e_first)
    e1.record()
    t_issued = time.perf_counter()
    # Poll for completion (a blocking synchronize wakes up tens of us late), then synchronize.  The poll also asks the stream for its
    # status: the runtime then retires finished launches while the host has nothing else to do, instead of all K of them inside
    # torch.cuda.synchronize() (13-22 us after the work was known to be complete, 4-5 us this way: profiles/r04_k20_timeline.txt).
    while not e1.query():
        if stream_query is not None:
            stream_query(stream)
    t_seen = time.perf_counter()
    wall = region.close()
    if os.environ.get("SKR_BENCH_TIMELINE"):  # where the wall clock of a short window goes, host side (us from t0)
        print(f"[timeline] launches issued {1e6 * (t_issued - t0):.1f} | completion seen {1e6 * (t_seen - t0):.1f} | synchronized {1e6 * wall:.1f} | event span {1e3 * e0.elapsed_time(e1):.1f}", file=sys.stderr)
    # HIP events on the launch stream.  PRIMARY clock of the roofline keys: the whole timed region, e0 -> e1 over all K steps
    # (the stream is empty when the region starts, so the first launch's cold-queue dispatch latency is inside it, as it is inside
    # `value` / ms_per_step).  Secondary (roofline.steady_state): steps 2..K, back to back behind the first one -- the number
    # rocprofv3's per-kernel average agrees with; K = 1 has no such window.
    span_ms = e0.elapsed_time(e1) / args.steps
    steady_ms = e_first.elapsed_time(e1) / (args.steps - marked[0]) if args.steps > marked[0] > 0 else span_ms
    # configs with a noise generator: a second region with the step launches alone (noise tensors already resident), after the
    # contract's region -- the 8(d) roofline of those configs is stated for the step kernel, "generator reported separately"
    kernels_ms = None
    if generator is not None:
        run(max(args.warmup, 2), draw=False)
        f0.record()
        run(args.steps, offset=args.warmup, draw=False)
        f1.record()
        torch.cuda.synchronize(dev)
        kernels_ms = f0.elapsed_time(f1) / args.steps

    coll_dev = dev if backend == "nccl" else None
    ranks = rank_spread({"wall_us_per_step": wall * 1e6 / args.steps, "span_us_per_step": span_ms * 1e3, "steady_us_per_step": steady_ms * 1e3}, dist, coll_dev)
    vals = max_over_ranks([wall, span_ms, steady_ms, kernels_ms or 0.0], dist, coll_dev)  # the slowest rank defines the step time
    wall, span_ms, steady_ms, kernels_ms = vals[0], vals[1], vals[2], (vals[3] if generator is not None else None)

    extras = rank == 0 and not args.no_extras
    wrapper = wrapper_rate(wl, batch, dev, shard) if extras else None

    # streaming reference point of this box: runtime device-to-device copy of 512 MiB (read + write
    # counted), far larger than the 256 MiB Infinity Cache -- context for the roofline fraction, not a target
    copy_gbs = None
    if extras:
        src = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record()
        torch.cuda.synchronize(dev)
        copy_gbs = 10 * 2 * src.numel() / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        steps_per_s = aggregate_rate(args.steps, world, wall)
        algo_bytes = numel * wl.bytes_per_elem
        roof_ms = kernels_ms if kernels_ms is not None else span_ms  # the step kernels' own time
        achieved = algo_bytes / (roof_ms * 1e-3) / 1e9
        steady = algo_bytes / (steady_ms * 1e-3) / 1e9
        if args.no_traffic:
            traffic, traffic_source, breakdown = None, "skipped (--no-traffic)", {}
        else:
            traffic, traffic_source, breakdown = measure_traffic(wl, batch, steps=40 if not heavy else 12)
        if traffic is None:
            committed, where = load_traffic(wl)
            traffic, traffic_source = committed, f"{where}; live: {traffic_source}"
        step_needle = wl.pmc_kernels[0][0]
        roofline = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "clock": ("HIP events around all K solver steps on the launch stream (cold-queue start of the region included)" if kernels_ms is None else
                      "HIP events around K solver steps' STEP launches alone (noise tensors resident), a second region behind the contract's; SURVEY 8(d) states this config's bytes for the step kernel, generator separately (see whole_step)"),
            "us_per_step": roof_ms * 1e3,
            "us_per_launch": roof_ms * 1e3 / launches_per_step,
            "launches_per_step": launches_per_step,
            "achievable_peak": HBM_ACHIEVABLE_GBS,  # guides/MI355X_MICROARCH.md: ~6.3 TB/s achievable of the 8 TB/s spec
            "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
            "steady_state": {  # steps 2..K only (what rocprofv3's per-kernel average of the same command shows)
                "us_per_step": steady_ms * 1e3,
                **({"achieved": steady, "frac": steady / HBM_PEAK_GBS} if kernels_ms is None else {}),
            },
            "wall_clock": {  # the host clock `value` is computed from: region start/stop cost spread over K steps
                "us_per_step": wall * 1e6 / args.steps,
                "frac": algo_bytes * args.steps / wall / 1e9 / HBM_PEAK_GBS,
            },
            "traffic": breakdown.get(step_needle, traffic) if breakdown else traffic,
            "traffic_source": traffic_source,
            # the same time priced on the bytes the step kernels actually moved (PMC): differs from `frac` where the implementation moves
            # fewer bytes than SURVEY 8(d) counts (cfg5: stored derivatives are read as 2 B, 8(d) counts verbatim (x, out) pairs at 4 B)
            "frac_on_measured_traffic": None,
            "algorithmic_bytes_per_step": algo_bytes,
            "algorithmic_bytes_per_element": wl.bytes_per_elem,
            "kernel": wl.kernel,
            "measured_d2d_memcpy": copy_gbs,  # hipMemcpy D2D of 512 MiB on this box, read+write GB/s
            # every rank's own clocks (min / max / per rank): each wall is read after that rank's own synchronise, before the closing barrier
            "ranks": ranks,
        }
        if roofline["traffic"]:
            roofline["frac_on_measured_traffic"] = roofline["traffic"] / (roof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if launches_per_step == 1:
            roofline["algorithmic_bytes_per_launch"] = algo_bytes
        if wl.mix_ceiling:
            roofline["mix_ceiling"] = wl.mix_ceiling  # a no-arithmetic kernel with this launch's traffic mix, same buffers
        if kernels_ms is not None:
            roofline["whole_step"] = {  # generator launches + step launches: what `value` counts
                "us_per_step": span_ms * 1e3,
                "generator_us_per_step": (span_ms - kernels_ms) * 1e3,
                "frac_of_step_bytes": algo_bytes / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic_all_kernels": traffic,
                "traffic_by_kernel": breakdown or None,
                "generator_note": wl.generator_bytes_note,
            }
        out = {
            "metric": wl.metric,
            "value": steps_per_s,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl.title.format(batch=batch),
                "name": wl.name,
                "global_batch": batch * world,
                "per_gpu_batch": batch,
                "latent_dtype": "bf16",
                "compute_dtype": "f32 registers, fp64 host coefficients",
                "schedule_steps": wl.schedule_steps,
                "steady_state_steps": [wl.steady[0], wl.steady[-1]],
                "launches_per_step": launches_per_step + (sum(n for _, n in wl.pmc_kernels[1:]) if generator is not None else 0),
                "buffer_sets": nsets,
                "buffer_set_bytes": set_bytes,
                "precondition_steps": precondition,
                "parallelism": f"batch-shard x{world}, no collective",
            },
            "roofline": roofline,
            "wrapper_steps_per_s": wrapper,
        }
        if wl.name == "headline":
            out["graph_loop_cfg2"] = None if args.no_extras else graph_loop_rate(dev)
        if not args.no_cpu_baseline:
            # host-only work: at N > 1 rank 0 times it (shorter) while the other ranks wait at the closing barrier
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds if world == 1 else min(args.cpu_seconds, 8.0), world)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()  # rank 0's informational extras (wrapper rate, graph loop, counters, CPU baseline) are done: every rank leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
