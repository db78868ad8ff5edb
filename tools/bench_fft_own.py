"""Colored draws of shapes the LDS plane kernels do not take (large odd factors, odd sides, widths not divisible by 4): the library's own
any-length transforms (skr_fft_own.hip, Bluestein on the LDS tile transform) against hipFFT on the same seeds.
usage: python tools/bench_fft_own.py   (one GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd import _hip
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step

lib = _hip.load()
shapes = [((16, 60, 104), 64), ((16, 90, 160), 32), ((4, 152, 104), 64), ((4, 97, 97), 64), ((4, 30, 90), 256), ((16, 13, 60, 104), 8), ((3, 250, 250), 16), ((4, 720, 1280), 2), ((4, 100), 1024), ((16, 66, 130), 64), ((4, 154, 182), 64), ((2, 2002), 256)]
print(f"{'unit':>22s} {'B':>5s} {'own us':>10s} {'hipFFT us':>10s} {'ratio':>6s}  rel. difference of the two results")
for unit, B in shapes:
    res, t = {}, {}
    for name, flag in (("own", 0), ("hipfft", 1)):
        assert lib.skr_set_tuning(b"hipfft", flag) == 0
        try:
            g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(B)), props=PN.ColoredProps(), dtype=torch.float32)
            for _ in range(3):
                out = g.generate(Step(0.45, 0.5))
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                g.generate(Step(0.45, 0.5))
            ev[1].record()
            torch.cuda.synchronize()
            t[name], res[name] = ev[0].elapsed_time(ev[1]) / 10 * 1e3, out
        finally:
            lib.skr_set_tuning(b"hipfft", -1)
    d = ((res["own"].double() - res["hipfft"].double()).abs().max() / res["hipfft"].double().abs().max()).item()
    print(f"{str(unit):>22s} {B:5d} {t['own']:10.1f} {t['hipfft']:10.1f} {t['own'] / t['hipfft']:6.2f}  {d:.2e}   (own transforms run so far: {lib.skr_stat(b'own_fft_execs')}, hipFFT: {lib.skr_stat(b'hipfft_execs')})")
