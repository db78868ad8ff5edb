"""GPU box: run the seeded random parity sweeps of tests/test_step_gpu.py over many more seeds than the test-suite does
(usage: python tests/soak.py [first_seed last_seed]).  Found the fp16 fused multiply-convert double-rounding mismatch."""
import sys, os, traceback
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import torch
import test_step_gpu as T
dev = torch.device("cuda:0")
from skrample_amd import _hip; _hip.load()
bad = 0
import random
import numpy as np
from test_step_gpu import PD, OW, OA, ON, SAMPLERS, SCHEDULES, MODELS, oracle_schedule, assert_close, FLOW_SCHEDULES, VP_SCHEDULES


def sweep_float64(seed, dev):
    "compute_scale = float64 (fp64 accumulate kernels), fp32 / fp64 tensors, injected noise; bar 1e-10 for fp64 tensors"
    rng = random.Random(4000 + seed)
    sampler = rng.choice(sorted(SAMPLERS))
    mk_o, mk_p = SAMPLERS[sampler]
    if rng.random() < 0.4:
        sname, mname = rng.choice(FLOW_SCHEDULES), rng.choice(("flow", "data", "v"))
    else:
        sname = rng.choice(VP_SCHEDULES[:5])
        mname = rng.choice(("eps", "v", "data"))
    dtype = rng.choice((torch.float32, torch.float64))
    shape = (rng.randint(1, 3), rng.randint(1, 5), rng.choice((8, 13, 16)), rng.choice((8, 10, 17)))
    steps = rng.randint(2, 9)
    g = torch.Generator().manual_seed(seed)
    w = PD.SkrampleWrapperScheduler(mk_p(), SCHEDULES[sname][1](), MODELS[mname][1], compute_scale=torch.float64)
    o = OW.StepDriver(mk_o(), oracle_schedule(sname, steps), MODELS[mname][0], compute=torch.float64)
    w.set_timesteps(steps); o.set_timesteps(steps)
    noises = [torch.randn(shape, generator=g, dtype=torch.float64) for _ in range(steps)]
    w._noise_generator = T.Injected(noises, dev)
    x = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
    for i, t in enumerate(w.timesteps):
        out = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
        try:
            ref = o.step(out, t, x, noise=noises[i])[0]
        except ZeroDivisionError:
            return
        got = w.step(out.to(dev), t, x.to(dev), return_dict=False)[0]
        if not torch.isfinite(ref).all():
            return
        assert got.dtype == dtype
        err = T.rel_err(got, ref)
        assert err <= (1e-10 if dtype == torch.float64 else 1e-6), (f"{sampler}/{sname}/{mname}/{dtype}/{shape}/{steps} step {i}", err)
        x = ref


def sweep_in_kernel_philox(seed, dev):
    "like test_random_sweep_vs_oracle, but the noise is drawn by the engine (in-kernel Philox or skr_noise_random) from per-sample seeds"
    rng = random.Random(9000 + seed)
    sampler = rng.choice(sorted(SAMPLERS))
    mk_o, mk_p = SAMPLERS[sampler]
    if not OA.require_noise(mk_o()):
        return
    if rng.random() < 0.4:
        sname, mname = rng.choice(FLOW_SCHEDULES), rng.choice(("flow", "data", "v"))
    else:
        sname = rng.choice(VP_SCHEDULES[:5])
        mname = rng.choice(("eps", "v", "data"))
    dtype = rng.choice((torch.float32, torch.bfloat16, torch.float16))
    shape = (rng.randint(1, 3), rng.randint(1, 5), rng.choice((8, 13, 16, 32)), rng.choice((8, 10, 16, 17, 64)))
    steps = rng.randint(2, 9)
    seeds = [rng.randrange(2**64) for _ in range(shape[0])]
    g = torch.Generator().manual_seed(seed)
    w = PD.SkrampleWrapperScheduler(mk_p(), SCHEDULES[sname][1](), MODELS[mname][1])
    o = OW.StepDriver(mk_o(), oracle_schedule(sname, steps), MODELS[mname][0])
    w.set_timesteps(steps); o.set_timesteps(steps)
    n = int(np.prod(shape[1:]))
    x = torch.randn(shape, generator=g).to(dtype)
    for i, t in enumerate(w.timesteps):
        out = torch.randn(shape, generator=g).to(dtype)
        noise = torch.from_numpy(np.stack([ON.philox_normal(s, i * 256, n) for s in seeds])).reshape(shape)
        try:
            ref = o.step(out, t, x, noise=noise)[0]
        except ZeroDivisionError:
            return
        got = w.step(out.to(dev), t, x.to(dev), generator=seeds, return_dict=False)[0]
        if not torch.isfinite(ref.float()).all():
            return
        assert_close(got, ref, dtype, f"{sampler}/{sname}/{mname}/{dtype}/{shape}/{steps} step {i}", flips=0.10)
        x = ref


lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (48, 700)
for seed in range(lo, hi):
    for fn in (T.test_random_sweep_vs_oracle, T.test_random_sweep_runge_kutta_vs_oracle, sweep_in_kernel_philox, sweep_float64, T.chunked_sweep_case):
        try:
            fn.__wrapped__(seed, dev) if hasattr(fn, "__wrapped__") else fn(seed, dev)
        except Exception as e:
            bad += 1
            print("FAIL", fn.__name__, seed, type(e).__name__, str(e)[:300])
print("done, failures:", bad)
