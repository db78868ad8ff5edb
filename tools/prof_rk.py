"""cfg5 shard (RKUltra order 6 = Cash-Karp, eps, Scaled, Pyramid noise, 64x4x256x256 bf16) for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.pytorch import noise as PN
dev = torch.device("cuda:0")
shape = (64, 4, 256, 256)
w = PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps())
g = torch.Generator(device=dev).manual_seed(0)
outs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(4)]
x0 = torch.randn(shape, device=dev, generator=g).bfloat16()
seeds = list(range(shape[0]))
for rep in range(3):
    w.set_timesteps(4)
    x = x0
    for i, t in enumerate(w.timesteps):
        x = w.step(outs[i % 4], t, x, generator=seeds, return_dict=False)[0]
torch.cuda.synchronize()
