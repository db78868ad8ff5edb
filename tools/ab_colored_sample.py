"""Colored draws of 3-D power-of-two units: the one-launch kernel (`colored_sample`: the planes of a sample meet inside the kernel)
against the three launches (plane kernel, outer axis, plane kernel; SKR_FFT_NO_SAMPLE=1) -- values and time, alternating in one process.
The one-launch kernel is an experiment that is NOT in the shipped library (it measured slower, see skr_colored.hip): build a variant with

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSKR_COLORED_SAMPLE -o tools/tune/libskrample_hip_sample.so skrample_amd/csrc/*.hip

and run  SKR_LIB=tools/tune/libskrample_hip_sample.so python tools/ab_colored_sample.py  (-> profiles/r03_colored_one_launch_experiment.txt)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd import _hip
if os.environ.get("SKR_LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["SKR_LIB"])
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step


def draw_time(g, st, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        g.generate(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for batch, unit, dtype in ((256, (16, 128, 128), torch.bfloat16), (64, (4, 128, 128), torch.bfloat16), (256, (4, 64, 64), torch.bfloat16), (32, (8, 128, 128), torch.float32),
                           (64, (2, 64, 64), torch.float16), (3, (16, 64, 64), torch.float32), (1024, (4, 128, 128), torch.bfloat16)):
    st = Step(0.45, 0.5)
    res, times = {}, {True: [], False: []}
    gens = {}
    for one in (True, False):
        os.environ.pop("SKR_FFT_NO_SAMPLE", None)
        if not one:
            os.environ["SKR_FFT_NO_SAMPLE"] = "1"
        gens[one] = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(batch)), props=PN.ColoredProps(), dtype=dtype)
        res[one] = gens[one].generate(st).float().cpu()
        gens[one].generate(st)
    for rnd in range(6):
        for one in ((True, False) if rnd % 2 == 0 else (False, True)):
            os.environ.pop("SKR_FFT_NO_SAMPLE", None)
            if not one:
                os.environ["SKR_FFT_NO_SAMPLE"] = "1"
            times[one].append(draw_time(gens[one], st, 20))
    os.environ.pop("SKR_FFT_NO_SAMPLE", None)
    diff = (res[True] - res[False]).abs().max().item() / res[False].abs().max().item()
    same = (res[True] == res[False]).float().mean().item()
    print(f"B={batch} {unit} {str(dtype)[6:]}: one launch {statistics.median(times[True]):7.1f} us  three launches {statistics.median(times[False]):7.1f} us   max rel diff {diff:.2e}  identical {same:.4f}  std {res[True].std().item():.4f}", flush=True)
