import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT
dev = torch.device("cuda:0")
shape, steps = (2, 4, 32, 32), 400
xs = [torch.randn(shape, device=dev).bfloat16() for _ in range(6)]
outs = [torch.randn(shape, device=dev).bfloat16() for _ in range(6)]
w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()))
for rep in range(2):
    w.set_timesteps(steps); ts = w.timesteps.tolist()
    for i, t in enumerate(ts):
        w.step(outs[i % 6], t, xs[i % 6], generator=[1, 2], return_dict=False)
w.set_timesteps(steps); ts = w.timesteps.tolist()
w.step(outs[0], ts[0], xs[0], generator=[1, 2], return_dict=False)
pr = cProfile.Profile(); pr.enable()
for i, t in enumerate(ts[1:], 1):
    w.step(outs[i % 6], t, xs[i % 6], generator=[1, 2], return_dict=False)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
