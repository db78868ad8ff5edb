"""cfg2 (B=64) sampler loop: eager wrapper vs whole-loop HIP graph (no network: the model call returns a fixed tensor)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT
from skrample_amd.graphs import capture_sampling_loop
dev = torch.device("cuda:0")
for B in (64, 256):
    shape, steps = (B, 4, 128, 128), 20
    x0 = torch.randn(shape, device=dev).bfloat16(); fixed = torch.randn(shape, device=dev).bfloat16()
    pool = [fixed] + [torch.randn(shape, device=dev).bfloat16() for _ in range(3)]; n_calls = [0]
    def net(x, t):
        n_calls[0] += 1
        return pool[n_calls[0] % 4]  # distinct buffers in turn: the wrapper guards its aliased history
    seeds = list(range(B))
    w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()))
    def eager():
        w.set_timesteps(steps); x = x0
        for t in w.timesteps.tolist(): x = w.step(net(x, t), t, x, generator=seeds, return_dict=False)[0]
        return x
    for _ in range(3): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
    loop = capture_sampling_loop(PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())), net, x0, steps, seeds=seeds)
    for _ in range(3): loop.graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): loop.graph.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
    print(f"B={B}: eager {te/steps*1e6:.1f} us/step ({steps/te:.0f} steps/s)   graph {tg/steps*1e6:.1f} us/step ({steps/tg:.0f} steps/s)")
