"""Colored draws of units whose leading axes are direct DFTs (a channel axis that is not a power of two <= 16; video latents: channels x
frames x height x width): the last two axes on the LDS plane kernels against hipFFT for the last three (SKR_FFT_NO_PLANES=1).  The library
routes only 3-D units with small odd factors to the plane kernels (where they measured faster); the 4-axis rows show hipFFT on both sides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step

for batch, unit in ((2, (16, 13, 60, 104)), (2, (16, 21, 64, 64)), (8, (16, 21, 64, 64)), (4, (16, 8, 96, 96)), (8, (4, 5, 96, 96)), (32, (4, 5, 96, 96)), (4, (16, 16, 128, 128)), (64, (3, 96, 96)), (64, (12, 64, 64)), (16, (5, 128, 128)), (1, (16, 21, 90, 160))):
    row = []
    outs = []
    for planes in (True, False):
        os.environ.pop("SKR_FFT_NO_PLANES", None)
        if not planes:
            os.environ["SKR_FFT_NO_PLANES"] = "1"
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
    os.environ.pop("SKR_FFT_NO_PLANES", None)
    n = batch
    for d in unit:
        n *= d
    diff = (outs[0] - outs[1]).abs().max().item()
    print(f"Colored B={batch} {unit}: plane kernels {row[0]:.3f} ms ({n / row[0] / 1e6:.1f} Gelem/s)   hipFFT {row[1]:.3f} ms   max abs diff {diff:.3g}", flush=True)
