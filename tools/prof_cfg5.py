"""BASELINE config 5 shard (RKUltra-6 SDE + Pyramid, 64x4x256x256 bf16): a few steps through the wrapper, for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.pytorch import noise as PN
dev = torch.device("cuda:0")
shape = (64, 4, 256, 256)
w = PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps(), prefetch_noise=os.environ.get("SKR_PREFETCH", "1") == "1")
g = torch.Generator(device=dev).manual_seed(0)
outs = [torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(4)]
x0 = torch.randn(shape, device=dev, generator=g).bfloat16()
seeds = list(range(shape[0]))
for rep in range(3):
    w.set_timesteps(4)
    ts = list(w.timesteps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = x0
    for i, t in enumerate(ts):
        x = w.step(outs[i % 4], t, x, generator=seeds, return_dict=False)[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"cfg5 shard: {dt / len(ts) * 1e6:.1f} us per stage call, {dt / 4 * 1e6:.1f} us per step ({len(ts)} calls)")
