#!/usr/bin/env python3
"""Headline benchmark: fused DPM-2 SDE sampler step (eps-prediction, Karras(Scaled) sigmas) over
B x 4 x 128 x 128 bf16 latents -- BASELINE.json `metric`, north-star shape B = 256 per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one solver step over one B=256 batch resident in HBM: exactly one launch of the fused
kernel through the C ABI (skr_step_launch).  The launch plans are not hand-written here: the real
scheduler wrapper (skrample_amd.diffusers.SkrampleWrapperScheduler) is run once over a 20-step
schedule with launch tracing on, and the plans it emitted for the steady-state steps 5..14 are
replayed on >= 4 rotating buffer sets (footprint > 256 MB Infinity Cache).  Multi-GPU = batch shards
with no collective (weak scaling, B = 256 per GPU); noise seeds are indexed by global sample id.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

B_PER_GPU, C, H, W = 256, 4, 128, 128
SCHEDULE_STEPS = 20
STEADY = list(range(5, 15))  # steady-state (order-2) step indices that are cycled
ALGO_BYTES_PER_ELEM = 10  # x 2 + model_out 2 + history pair 4 + y 2 (in-kernel noise 0)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md:36)
HBM_ACHIEVABLE_GBS = 6300.0  # what that guide measures as achievable on this part (float4 copy, "8 TB/s peak (spec); ~6.3 TB/s achievable")


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="samples per GPU (default: the north-star 256)")
    ap.add_argument("--sets", type=int, default=6, help="rotating buffer sets (>= 4)")
    ap.add_argument("--precondition", type=int, default=1000, help="untimed conditioning launches before the warm-up (see the comment at its use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 FETCH_SIZE / WRITE_SIZE passes (roofline.traffic falls back to the committed summary)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="diagnostic: kernel-selection switches (skr_set_tuning) applied before anything is launched")
    ap.add_argument("--drift", type=int, default=0, help="diagnostic: run this many launches back to back and print the average launch time of every block of 50 (then exit)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no wrapper-rate and graph-loop keys (used by the counter passes)")
    return ap.parse_args()


def make_wrapper():
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.sampling import structured as PT

    # alias_history=True: the synthetic "network" hands over fresh tensors, which the default ("auto") would find out by its second
    # call -- stated here so that the traced plans name the caller's own tensors from the first step on (capture_plans maps pointers)
    w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), alias_history=True)
    w.set_timesteps(SCHEDULE_STEPS)
    return w


def capture_plans(dev: torch.device, shard):
    """Run the real wrapper once with tracing and return, per steady-state step, the emitted plan and
    the role of each input pointer (x, out, x_prev, out_prev)."""
    from skrample_amd import _hip

    w = make_wrapper()
    batch = shard.batch
    g = torch.Generator(device=dev).manual_seed(shard.input_seed())
    shape = (batch, C, H, W)
    gens = shard.seeds()  # per-sample seeds by GLOBAL sample index (skrample_amd/sharding.py)
    x = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
    plans = {}
    prev_pair = None
    _hip.trace = []
    try:
        for i, t in enumerate(w.timesteps):
            out = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
            _hip.trace.clear()
            # seeds are plain ints (the generator protocol only needs initial_seed())
            nxt = w.step(out, t, x, generator=[_Seed(s) for s in gens], return_dict=False)[0]
            assert len(_hip.trace) == 1, "a solver step must be exactly one fused launch"
            plan, inputs, _, _, seeds, numel = _hip.trace[0]
            roles = []
            for tin in inputs:
                if tin.data_ptr() == x.data_ptr():
                    roles.append("x")
                elif tin.data_ptr() == out.data_ptr():
                    roles.append("out")
                elif prev_pair and tin.data_ptr() == prev_pair[0].data_ptr():
                    roles.append("x_prev")
                elif prev_pair and tin.data_ptr() == prev_pair[1].data_ptr():
                    roles.append("out_prev")
                else:
                    raise RuntimeError("unexpected operand in the traced launch")
            plans[i] = (plan, roles, seeds, numel)
            prev_pair = (x, out)
            x = nxt
    finally:
        _hip.trace = None
    torch.cuda.synchronize(dev)
    return plans


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


class _Seed:
    "minimal stand-in for torch.Generator as a seed carrier (initial_seed only)"

    def __init__(self, s: int):
        self.s = s

    def initial_seed(self) -> int:
        return self.s


def _cpu_port_rate(seconds: float, threads: int) -> tuple[float, int, int]:
    "B=256-equivalent steps/s of the oracle's reference-order port with `threads` torch threads"
    from skr_oracle import samplers as OA
    from skr_oracle import schedules as OS
    from skr_oracle import wrapper as OW

    torch.set_num_threads(threads)
    sub = CPU_SAMPLE  # cfg2's own batch; the port's cost is linear in the batch (per-sample generators, elementwise passes)
    drv = OW.StepDriver(OA.make("dpm", 2, eta=1), OS.karras(OS.scaled(), steps=SCHEDULE_STEPS), "eps", mimic_copies=True)
    g = torch.Generator().manual_seed(1234)
    seeds = [42 + i for i in range(sub)]
    times: list[float] = []
    t_start = time.perf_counter()
    schedules = 0
    while time.perf_counter() - t_start < seconds:  # whole 20-step schedules until ~`seconds` of CPU work
        drv.set_timesteps(SCHEDULE_STEPS)
        x = torch.randn(sub, C, H, W, generator=g).bfloat16()
        for i, t in enumerate(drv.timesteps):
            out = torch.randn(sub, C, H, W, generator=g).bfloat16()
            t0 = time.perf_counter()
            x = drv.step(out, t, x, seeds=seeds)[0]
            dt = time.perf_counter() - t0
            if i in STEADY:
                times.append(dt)
        schedules += 1
    per_step_full = (sum(times) / len(times)) * (B_PER_GPU / sub)
    return 1.0 / per_step_full, len(times), schedules


CPU_SAMPLE = 64


def cpu_baseline(seconds: float) -> dict:
    """The oracle (reference-order torch CPU port, incl. the reference's deep copies and per-sample randn +
    stack) timed on this box's host cores on a bounded sample: a slice of the batch, scaled to B=256.
    Timed twice -- with torch's default thread count and with 16 threads (elementwise passes over a few MB do not
    scale to a whole socket) -- and the faster of the two is reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    default_threads = torch.get_num_threads()
    trials = {}
    for threads in sorted({default_threads, min(16, default_threads)}):
        trials[threads] = _cpu_port_rate(seconds / 2, threads)
    torch.set_num_threads(default_threads)
    best = max(trials, key=lambda k: trials[k][0])
    rate, nsteps, schedules = trials[best]
    others = ", ".join(f"{k} threads: {v[0]:.2f} steps/s" for k, v in trials.items())
    return {
        "value": rate,
        "unit": "steps/s",
        "cores": best,
        "kind": "port",
        "sample": f"oracle StepDriver (reference op order, fp32 compute, per-sample randn+stack, deep copies) on {CPU_SAMPLE} of {B_PER_GPU} samples, "
        f"{nsteps} steady-state steps over {schedules} runs of a {SCHEDULE_STEPS}-step schedule, time scaled x{B_PER_GPU // CPU_SAMPLE} to B={B_PER_GPU}; "
        f"best of ({others})",
    }


def graph_loop_rate(dev: torch.device) -> dict | None:
    """SURVEY 8(f) rank 1, reported beside the headline: BASELINE config 2 itself (B=64x4x128x128, DPM-2 SDE, Karras, 20 steps)
    as ONE HIP graph of the whole sampler loop with device-resident step scalars (skr_step_launch_indexed) -- the launch-bound
    regime where the eager wrapper is host-limited.  Sampler only: the "network" hands back pre-generated tensors."""
    try:
        import skrample_amd.diffusers as PD
        import skrample_amd.scheduling as PS
        from skrample_amd.graphs import capture_sampling_loop
        from skrample_amd.sampling import structured as PT

        batch, steps = 64, SCHEDULE_STEPS
        shape = (batch, C, H, W)
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
            "workload": f"BASELINE config 2: DPM-2 SDE + Karras, eps-pred, B={batch}x{C}x{H}x{W} bf16, {steps}-step loop, sampler only",
            "graph_steps_per_s": steps / graph_s,
            "graph_us_per_step": graph_s / steps * 1e6,
            "eager_steps_per_s": steps / eager_s,
            "eager_us_per_step": eager_s / steps * 1e6,
            "mode": "one HIP graph per loop, step scalars read from a device-resident table (skr_step_launch_indexed)",
        }
    except Exception as exc:  # informational key: never take the headline down
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


def load_traffic() -> float | None:
    """HBM bytes per launch of the headline kernel from the committed PMC passes of this same command
    (profiles/r03_pmc_traffic.json, else round 2's: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per the guide): the
    fallback when the live passes below cannot run"""
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    if not os.path.isfile(path):
        path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    try:
        return float(json.load(open(path))["hbm_bytes_per_launch"])
    except Exception:
        return None


HEADLINE_KERNEL = "step_kernel_k1<skr::bf16_t, 4, true"  # K=4 bf16 operands + in-kernel Philox (DPM-2 SDE steady state)


def measure_traffic(batch: int) -> tuple[float | None, str]:
    """HBM bytes per launch of the headline kernel, measured now: this command is re-run twice as a child process under
    `rocprofv3 --pmc <counter> --kernel-trace` (FETCH_SIZE and WRITE_SIZE in separate passes, as the guide's HBM section
    prescribes), the counter rows of the B=`batch` launches of the headline kernel are averaged, and the guide's gfx950
    corrections are applied (values are KiB; FETCH_SIZE counts the 128-B requests of wide coalesced streams as 64 B -> x2;
    WRITE_SIZE is exact).  Returns (bytes, provenance); (None, reason) when the profiler is not usable here."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile

    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    grid = batch * C * H * W // 8  # threads of one headline launch (8 elements each)
    tmp = tempfile.mkdtemp(prefix="skr_pmc_", dir="/tmp")
    means, counts = {}, {}
    try:
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, name)
            cmd = [exe, "--pmc", name, "--kernel-trace", "-d", out_dir, "-o", "prof", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "60", "--warmup", "5", "--precondition", "0",
                   "--batch", str(batch), "--no-cpu-baseline", "--no-traffic", "--no-extras"]  # fmt: skip
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
            env["TMPDIR"] = "/tmp"
            proc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=120)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if proc.returncode != 0 or not files:
                return None, f"{name} pass failed (rc {proc.returncode}): {proc.stderr.decode(errors='replace')[-160:]}"
            vals = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
                    if HEADLINE_KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name and int(r["Grid_Size"]) == grid]  # fmt: skip
            vals = [v for v in vals if v > 0.75 * max(vals)] if vals else vals  # steady-state launches (order-1 steps read no history)
            if len(vals) < 10:
                return None, f"{name} pass: only {len(vals)} headline launches found"
            means[name], counts[name] = statistics.mean(vals), len(vals)
    except subprocess.TimeoutExpired:
        return None, "counter pass timed out"
    except Exception as exc:
        return None, f"{type(exc).__name__}: {exc}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = 2.0 * means["FETCH_SIZE"] * 1024.0 + means["WRITE_SIZE"] * 1024.0
    return total, (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in child processes "
                   f"({counts['FETCH_SIZE']} / {counts['WRITE_SIZE']} launches; KiB, FETCH x2 on gfx950)")


def main() -> None:
    args = parse()
    from skrample_amd.sharding import BatchShard, aggregate_rate, max_over_ranks

    shard = BatchShard.from_env(args.batch)
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
    batch = args.batch
    numel = batch * C * H * W
    plans = capture_plans(dev, shard)

    # rotating buffer sets: (x, out, x_prev, out_prev, y) each 2 B/elem -> 168 MB per set at B=256.
    # In a sampling loop `x`, `x_prev` and `y` are tensors the engine itself allocated (step results), so they
    # come from its output allocator (lazy.empty_output: 4 KiB-staggered placement); `out` / `out_prev` stand
    # for the network's tensors and come straight from torch's allocator.
    from skrample_amd.sampling.lazy import empty_output

    nsets = max(args.sets, 4)
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    sets = []
    for _ in range(nsets):
        bufs = {}
        for r in ("x", "out", "x_prev", "out_prev", "y"):
            if r in ("out", "out_prev"):
                bufs[r] = torch.randn(numel, device=dev, generator=g).to(torch.bfloat16)
            else:
                bufs[r] = empty_output((numel,), torch.bfloat16, dev)
                if r != "y":
                    bufs[r].copy_(torch.randn(numel, device=dev, generator=g))
        sets.append(bufs)

    stream = torch.cuda.current_stream(dev).cuda_stream
    calls = []
    for k in range(len(STEADY) * nsets):
        plan, roles, seeds, n = plans[STEADY[k % len(STEADY)]]
        assert n == numel and roles == ["x", "out", "x_prev", "out_prev"], roles
        bufs = sets[k % nsets]
        ptrs = (ctypes.c_void_p * len(roles))(*[bufs[r].data_ptr() for r in roles])
        calls.append((ctypes.byref(plan), ptrs, bufs["y"].data_ptr(), seeds.data_ptr()))

    def run(count: int, offset: int = 0, mark=None) -> None:
        launch = lib.skr_step_launch
        ncalls = len(calls)
        for i in range(count):
            p, ptrs, y, sd = calls[(offset + i) % ncalls]
            status = launch(p, ptrs, y, None, sd, numel, stream)
            if status:
                _hip.check(status, "skr_step_launch")
            if i == 0 and mark is not None:
                mark.record()  # behind the first timed launch: the kernel clock excludes the cold-queue start of the region

    try:  # HIP events on the launch stream, without the system-scope fence of torch's events (see _HipEvent)
        e0, e1, e_first, e_warm = (_HipEvent(stream) for _ in range(4))
    except Exception:
        e0, e1, e_first, e_warm = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    for ev in (e0, e1, e_first, e_warm):  # (torch creates its HIP event at the first record: do that outside the timed region)
        ev.record()
    if args.drift:  # how the launch time moves over a long back-to-back run (clock / power management), 50 launches per reading
        marks = [_HipEvent(stream) for _ in range(args.drift // 50 + 1)]
        torch.cuda.synchronize(dev)
        marks[0].record()
        for b in range(1, len(marks)):
            run(50, offset=50 * (b - 1))
            marks[b].record()
        torch.cuda.synchronize(dev)
        per = [marks[b - 1].elapsed_time(marks[b]) * 1e3 / 50 for b in range(1, len(marks))]
        print("us per launch, blocks of 50:", " ".join(f"{v:.2f}" for v in per))
        tail = per[len(per) // 2 :]
        print(f"second half of the run: {sum(tail) / len(tail):.3f} us per launch  (--tune {args.tune})")
        return
    # conditioning (untimed, before the contract's W warm-up steps): ~26 ms of back-to-back launches so that clocks and the page
    # tables of all buffer sets are in their steady state however small W and K are.  `--drift 4000` shows why it has to be this
    # long: from a cold start the launch time rises to 26.3-26.4 us between launches ~100 and ~350 (power management settling),
    # comes back by launch ~600 and then stays at 25.7-25.85 us for as long as the run lasts; a K=20 window opened after 300
    # launches sat in that transient.  The count is reported in the JSON line (config.precondition_launches).
    run(args.precondition)
    torch.cuda.synchronize(dev)
    run(args.warmup)
    e_warm.record()
    while not e_warm.query():  # poll, then synchronize: a blocking wait returns tens of us late, and the GPU would sit idle
        pass                   # (and start the timed region from a colder state) for that long
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    e0.record()
    run(args.steps, offset=args.warmup, mark=e_first)
    e1.record()
    while not e1.query():  # poll for completion (a blocking synchronize wakes up tens of us late), then synchronize
        pass
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    # HIP events on the launch stream.  PRIMARY clock of the roofline keys: the whole timed region, e0 -> e1 over all K launches
    # (the stream is empty when the region starts, so the first launch's cold-queue dispatch latency is inside it, as it is inside
    # `value` / ms_per_step).  Secondary (roofline.steady_state): launches 2..K, back to back behind the first one -- the number
    # rocprofv3's per-kernel average agrees with; K = 1 has no such window.
    span_ms = e0.elapsed_time(e1) / args.steps
    steady_ms = e_first.elapsed_time(e1) / (args.steps - 1) if args.steps > 1 else span_ms

    wall, span_ms, steady_ms = max_over_ranks([wall, span_ms, steady_ms], dist, dev if backend == "nccl" else None)  # the slowest rank defines the step time

    # wrapper-level rate (Python scheduler overhead included), for information
    wrapper_rate = None
    if rank == 0 and not args.no_extras:
        w = make_wrapper()
        shape = (batch, C, H, W)
        xs = [s["x"].view(shape) for s in sets]
        outs = [s["out"].view(shape) for s in sets]
        seeds = [_Seed(sd) for sd in shard.seeds()]
        for rep in range(3):
            w.set_timesteps(SCHEDULE_STEPS)
            ts = w.timesteps.tolist()
            w.step(outs[0], ts[0], xs[0], generator=seeds, return_dict=False)  # first step of a run also builds the per-sample generators
            torch.cuda.synchronize(dev)
            tw = time.perf_counter()
            for i, t in enumerate(ts[1:], start=1):
                w.step(outs[i % nsets], t, xs[i % nsets], generator=seeds, return_dict=False)
            torch.cuda.synchronize(dev)
            wrapper_rate = (SCHEDULE_STEPS - 1) / (time.perf_counter() - tw)

    # streaming reference point of this box: runtime device-to-device copy of 512 MiB (read + write
    # counted), far larger than the 256 MiB Infinity Cache -- context for the roofline fraction, not a target
    copy_gbs = None
    if rank == 0:
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
        algo_bytes = numel * ALGO_BYTES_PER_ELEM
        achieved = algo_bytes / (span_ms * 1e-3) / 1e9
        steady = algo_bytes / (steady_ms * 1e-3) / 1e9
        traffic, traffic_source = (None, "skipped (--no-traffic)") if args.no_traffic or world > 1 else measure_traffic(batch)
        if traffic is None:
            traffic, traffic_source = load_traffic(), f"profiles/r03_pmc_traffic.json (committed summary of the same passes; live: {traffic_source})"
        out = {
            "metric": "sampler steps/sec (fused DPM-2 SDE step, eps-pred, Karras sigmas, Bx4x128x128 bf16) + achieved HBM GB/s",
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
                "workload": f"DPM order-2 SDE (eta=1) + Karras(Scaled) sigmas, eps-pred, B={batch}x{C}x{H}x{W} bf16 latents per GPU "
                "(BASELINE north-star shape; cfg2 at 4x batch), in-kernel Philox noise, one fused launch per step",
                "global_batch": batch * world,
                "per_gpu_batch": batch,
                "latent_dtype": "bf16",
                "compute_dtype": "f32 registers, fp64 host coefficients",
                "schedule_steps": SCHEDULE_STEPS,
                "steady_state_steps": [STEADY[0], STEADY[-1]],
                "buffer_sets": nsets,
                "precondition_launches": args.precondition,
                "parallelism": f"batch-shard x{world}, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "clock": "HIP events e0 -> e1 around all K launches on the launch stream (cold-queue start of the region included)",
                "us_per_launch": span_ms * 1e3,
                "achievable_peak": HBM_ACHIEVABLE_GBS,  # guides/MI355X_MICROARCH.md: ~6.3 TB/s achievable of the 8 TB/s spec
                "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                "steady_state": {  # launches 2..K only (what rocprofv3's per-kernel average of the same command shows)
                    "us_per_launch": steady_ms * 1e3,
                    "achieved": steady,
                    "frac": steady / HBM_PEAK_GBS,
                },
                "wall_clock": {  # the host clock `value` is computed from: region start/stop cost spread over K launches
                    "us_per_step": wall * 1e6 / args.steps,
                    "frac": algo_bytes * args.steps / wall / 1e9 / HBM_PEAK_GBS,
                },
                "mix_ceiling": {  # a no-arithmetic kernel with this launch's traffic mix (4 x 16-byte reads + 1 write per lane), same buffers
                    "us_per_launch": 25.76, "frac": 0.814, "source": "profiles/r03_harness_lib_vs_ceilings.txt (kmix<R4,W1>, committed harness run)",
                },
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel": "skr::step_kernel_k1<bf16_t, K=4, NOISE=true> (one-trip, paced loads, XCD chunk map)",
                "measured_d2d_memcpy": copy_gbs,  # hipMemcpy D2D of 512 MiB on this box, read+write GB/s
            },
            "wrapper_steps_per_s": wrapper_rate,
            "graph_loop_cfg2": None if args.no_extras else graph_loop_rate(dev),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()  # rank 0's informational extras (wrapper rate, graph loop) are done: every rank leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
