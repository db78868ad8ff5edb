"""cfg3 (UniPC-3 SDE, flow, Linear, 256x16x128x128 bf16) for rocprofv3: Philox and tensor-noise variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT, models as PM
from skrample_amd.pytorch import noise as PN
dev = torch.device("cuda:0")
shape = (256, 16, 128, 128)
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(3)]
outs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(3)]
seeds = list(range(shape[0]))
for kind in (PN.Random, PN.Offset):
    w = PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=kind)
    for rep in range(2):
        w.set_timesteps(12)
        for i, t in enumerate(w.timesteps):
            w.step(outs[i % 3], t, xs[i % 3], generator=seeds, return_dict=False)
torch.cuda.synchronize()
