"""Colored draws on planes with large odd factors (60 x 104: 15 and 13; 90 x 160: 45 and 5) and video units built on them: the hand-written
route (mixed-radix plane kernels; for 4-axis units + the fused outer-axis pass) against hipFFT (SKR_FFT_NO_MIXED=1 SKR_FFT_NO_PLANES=1),
same seeds, with the hipFFT counters of each side (skr_stat) so that the route taken is on record.
Since the N-D transform became the library's own (round 4) colored_planes takes every plane it can hold; SKR_FFT_ODD_LIMIT=10 restores
the limit of rounds 3-4 (odd parts summing to more than 10 went to the N-D transform) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd import _hip
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step

lib = _hip.load()
for batch, unit in ((64, (4, 60, 104)), (64, (4, 90, 160)), (16, (16, 90, 160)), (2, (16, 13, 60, 104)), (8, (16, 13, 60, 104)), (1, (16, 21, 90, 160)), (4, (16, 21, 90, 160)), (64, (4, 120, 120)), (64, (4, 84, 84))):
    row, outs, used = [], [], []
    for hand in (True, False):
        for k in ("SKR_FFT_NO_PLANES", "SKR_FFT_NO_MIXED"):
            os.environ.pop(k, None)
            if not hand:
                os.environ[k] = "1"
        before = lib.skr_stat(b"hipfft_execs")
        g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(batch)), props=PN.ColoredProps(), dtype=torch.bfloat16)
        st = Step(0.45, 0.5)
        for _ in range(3):
            o = g.generate(st)
        outs.append(o.float().cpu())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            g.generate(st)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 10 * 1e3)
        used.append(lib.skr_stat(b"hipfft_execs") - before)
    for k in ("SKR_FFT_NO_PLANES", "SKR_FFT_NO_MIXED"):
        os.environ.pop(k, None)
    n = batch
    for d in unit:
        n *= d
    diff = (outs[0] - outs[1]).abs().max().item()
    print(f"Colored B={batch} {unit}: hand-written {row[0]:.3f} ms ({n / row[0] / 1e6:.1f} Gelem/s, {used[0]} hipFFT transforms)   hipFFT {row[1]:.3f} ms ({used[1]} transforms)   max abs diff {diff:.3g}", flush=True)
