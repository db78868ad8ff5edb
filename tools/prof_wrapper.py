"""Host cost of one eager scheduler step (cfg2 shape, GPU time is ~7 us so the loop is host-bound)."""
import sys, time, cProfile, pstats
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT
dev = torch.device('cuda:0')
w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
xs = [torch.randn(B,4,128,128, device=dev).bfloat16() for _ in range(4)]; outs = [torch.randn_like(xs[0]) for _ in range(4)]  # rotating buffers (the aliased history is guarded)
seeds = list(range(B))
def loop(n=3):
    for _ in range(n):
        w.set_timesteps(20)
        ts = w.timesteps.tolist()
        for i, t in enumerate(ts):
            w.step(outs[i % 4], t, xs[i % 4], generator=seeds, return_dict=False)
    torch.cuda.synchronize()
loop(2)
t=time.perf_counter(); loop(20); dt=time.perf_counter()-t; print("us/step", dt/400*1e6)
pr = cProfile.Profile(); pr.enable(); loop(20); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
