"""Per-config throughput of the BASELINE.json configs on one MI355X through the scheduler wrapper
(informational; the graded line is bench.py).  Algorithmic bytes per element per step follow SURVEY.md 8(d)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT, models as PM
from skrample_amd.pytorch import noise as PN

dev = torch.device("cuda:0")
CONFIGS = {
    "cfg2  DPM-2 SDE eps Karras   B=64x4x128x128 bf16": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())), (64, 4, 128, 128), 10, 1),
    "head  DPM-2 SDE eps Karras   B=256x4x128x128 bf16": (lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())), (256, 4, 128, 128), 10, 1),
    "cfg3  UniPC-3 SDE flow Linear+Philox B=256x16x128x128": (lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel()), (256, 16, 128, 128), 26, 1),
    "cfg3c UniPC-3 SDE flow Linear+Colored B=256x16x128x128": (lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps()), (256, 16, 128, 128), 30, 1),
    "cfg4  Adams-4 ODE v ZSNR     B=256x4x128x128 (1/8 of 2048)": (lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()), (256, 4, 128, 128), 18, 1),
    "cfg5  RKUltra-6 SDE+Pyramid  B=64x4x256x256 (1/8 of 512)": (lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps()), (64, 4, 256, 256), 100, 6),
}
NSETS = 4
for name, (mk, shape, bytes_per_elem, calls_per_step) in CONFIGS.items():
    w = mk()
    g = torch.Generator(device=dev).manual_seed(0)
    xs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(NSETS)]
    outs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(NSETS)]
    seeds = list(range(shape[0]))
    steps = 20 if calls_per_step == 1 else 4
    best = None
    for rep in range(3):
        w.set_timesteps(steps)
        ts = list(w.timesteps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x = xs[0]
        for i, t in enumerate(ts):
            x = w.step(outs[i % NSETS], t, x if calls_per_step > 1 else xs[i % NSETS], generator=seeds, return_dict=False)[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    n = 1
    for d in shape: n *= d
    per_call = best / len(ts)
    per_step = per_call * calls_per_step
    gbs = n * bytes_per_elem / per_step / 1e9
    print(f"{name:58s} {per_call*1e6:9.1f} us/call  {1/per_step:9.1f} steps/s  {gbs:7.0f} GB/s algorithmic ({gbs/8000:.2f} of 8 TB/s)")
