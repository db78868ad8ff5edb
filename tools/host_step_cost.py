"""Host cost of one eager scheduler step: tiny tensors (GPU time negligible), long schedule, first step excluded."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT
dev = torch.device("cuda:0")
shape, steps = (2, 4, 32, 32), 200
xs = [torch.randn(shape, device=dev).bfloat16() for _ in range(6)]
outs = [torch.randn(shape, device=dev).bfloat16() for _ in range(6)]
for name, mk in (("dpm2_sde", lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()))),
                 ("adams4", lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.Scaled())),
                 ("unipc3_sde", lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Scaled()))):
    w = mk()
    best = 1e9
    for rep in range(4):
        w.set_timesteps(steps)
        ts = w.timesteps.tolist()
        w.step(outs[0], ts[0], xs[0], generator=[1, 2], return_dict=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, t in enumerate(ts[1:], 1):
            w.step(outs[i % 6], t, xs[i % 6], generator=[1, 2], return_dict=False)
        dt = (time.perf_counter() - t0) / (steps - 1)
        torch.cuda.synchronize()
        best = min(best, dt)
    print(f"{name:12s} host {best * 1e6:6.2f} us/step")
    if name == "dpm2_sde" and len(sys.argv) > 1:
        w.set_timesteps(steps); ts = w.timesteps.tolist()
        w.step(outs[0], ts[0], xs[0], generator=[1, 2], return_dict=False)
        pr = cProfile.Profile(); pr.enable()
        for i, t in enumerate(ts[1:], 1):
            w.step(outs[i % 6], t, xs[i % 6], generator=[1, 2], return_dict=False)
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(16)
