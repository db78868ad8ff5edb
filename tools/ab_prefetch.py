"""A/B/A/B of `prefetch_noise` (next step's Pyramid / Offset noise drawn ahead on a side stream) on BASELINE config 5's shard and on a
DPM-2 SDE + Offset run: alternating runs in one process so that box drift cancels."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import models as PM
from skrample_amd.sampling import structured as PT
from skrample_amd.pytorch import noise as PN

dev = torch.device("cuda:0")


def timed(w, shape, calls_per_step, xs, outs, seeds):
    steps = 20 if calls_per_step == 1 else 4
    w.set_timesteps(steps)
    ts = list(w.timesteps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = xs[0]
    for i, t in enumerate(ts):
        x = w.step(outs[i % 4], t, x if calls_per_step > 1 else xs[i % 4], generator=seeds, return_dict=False)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(ts) * 1e6


def ab(name, mk, shape, calls_per_step):
    g = torch.Generator(device=dev).manual_seed(0)
    xs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(4)]
    outs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(4)]
    seeds = list(range(shape[0]))
    ws = {flag: mk(prefetch_noise=flag, alias_history=True) for flag in (True, False)}
    for flag in ws:  # warm: programs, workspaces
        for _ in range(3):
            timed(ws[flag], shape, calls_per_step, xs, outs, seeds)
    res = {True: [], False: []}
    for rnd in range(12):
        for flag in ((True, False) if rnd % 2 == 0 else (False, True)):
            res[flag].append(timed(ws[flag], shape, calls_per_step, xs, outs, seeds))
    for flag in (True, False):
        v = sorted(res[flag])
        print(f"{name:44s} prefetch_noise={str(flag):5s}  median {statistics.median(v):7.1f}  min {v[0]:7.1f}  max {v[-1]:7.1f} us/call", flush=True)


ab("cfg5 RKUltra-6 SDE + Pyramid 64x4x256x256", lambda **kw: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps(), **kw), (64, 4, 256, 256), 6)
ab("DPM-2 SDE + Pyramid 256x4x128x128", lambda **kw: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), noise_type=PN.Pyramid, noise_props=PN.PyramidProps(), **kw), (256, 4, 128, 128), 1)
ab("DPM-2 SDE + Offset 256x4x128x128", lambda **kw: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), noise_type=PN.Offset, noise_props=PN.OffsetProps(), **kw), (256, 4, 128, 128), 1)
ab("UniPC-3 SDE + Pyramid 256x16x128x128", lambda **kw: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Pyramid, noise_props=PN.PyramidProps(), **kw), (256, 16, 128, 128), 1)
ab("cfg3c UniPC-3 SDE + Colored 256x16x128x128", lambda **kw: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps(), **kw), (256, 16, 128, 128), 1)
ab("DPM-2 SDE + Colored 256x4x128x128", lambda **kw: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), noise_type=PN.Colored, noise_props=PN.ColoredProps(), **kw), (256, 4, 128, 128), 1)
