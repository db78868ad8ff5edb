"""Host cost of one replayed scheduler step, measured WITHOUT a GPU: tensors that report a HIP device (a torch.Tensor subclass
over CPU memory), the C-ABI launch replaced by a no-op.  Only the Python bookkeeping of SkrampleWrapperScheduler.step is
timed -- a development aid for the step-program fast path (the absolute numbers are this container's CPU, ~4x slower than
the GPU box's host; tools/prof_wrapper.py measures the real thing)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import skrample_amd.diffusers as PD
import skrample_amd.scheduling as PS
from skrample_amd import _hip
from skrample_amd.sampling import lazy, program
from skrample_amd.sampling import structured as PT


class Dev(torch.Tensor):
    @property
    def device(self):
        return torch.device("cuda", 0)

    @property
    def is_cuda(self):
        return True


def fake(shape, dtype=torch.bfloat16):
    return torch.zeros(shape, dtype=dtype).as_subclass(Dev)


class Lib:
    def __getattr__(self, name):
        return lambda *a, **k: 0


_hip.load = lambda: Lib()
_hip.step_launch_raw = lambda *a, **k: 0
_hip.current_stream_ptr = lambda device: 0
lazy.empty_output = lambda shape, dtype, device: fake(tuple(shape), dtype)
program.empty_output = lazy.empty_output
import skrample_amd.pytorch.noise as PN

PN.seeds_tensor = lambda values, device: torch.zeros(len(values), dtype=torch.int64).as_subclass(Dev)
PN.seed_device = lambda seed: torch.device("cuda", 0)

B = 64
shape = (B, 4, 16, 16)
w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), alias_history=True)
xs = [fake(shape) for _ in range(4)]
outs = [fake(shape) for _ in range(4)]
seeds = list(range(B))


def loop(n):
    for _ in range(n):
        w.set_timesteps(20)
        ts = w.timesteps.tolist()
        for i, t in enumerate(ts):
            w.step(outs[i % 4], t, xs[i % 4], generator=seeds, return_dict=False)


loop(3)
t = time.perf_counter()
loop(50)
print("us/step (host only, this CPU):", (time.perf_counter() - t) / 1000 * 1e6)
if len(sys.argv) > 1:
    pr = cProfile.Profile()
    pr.enable()
    loop(50)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
