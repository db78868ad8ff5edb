"""fp32 latents (256x4x128x128): Euler ODE, DPM-2 SDE, Adams-4, RKUltra-4 -- for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT, models as PM
dev = torch.device("cuda:0")
shape = (256, 4, 128, 128)
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(shape, device=dev, generator=g) for _ in range(4)]
outs = [torch.randn(shape, device=dev, generator=g) for _ in range(4)]
seeds = list(range(shape[0]))
for mk in (lambda: PD.SkrampleWrapperScheduler(PT.Euler(), PS.Scaled()),
           lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())),
           lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()),
           lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=4)):
    w = mk()
    for rep in range(2):
        w.set_timesteps(12)
        x = xs[0]
        for i, t in enumerate(w.timesteps):
            x = w.step(outs[i % 4], t, x if isinstance(w, PD.RKWrapperCore) else xs[i % 4], generator=seeds, return_dict=False)[0]
torch.cuda.synchronize()
