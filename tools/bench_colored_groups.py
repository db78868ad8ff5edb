"""Colored draw at the cfg3 shape (256 x (16,128,128) bf16) against SKR_COLORED_GROUP_MB (samples per pass through the three kernels).
usage: for m in 0 136 68; do SKR_COLORED_GROUP_MB=$m python tools/bench_colored_groups.py; done"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step

shapes = [((16, 128, 128), 256), ((4, 256, 256), 64), ((16, 96, 96), 256), ((16, 64, 64), 1024)]
for unit, B in shapes:
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(B)), props=PN.ColoredProps(), dtype=torch.bfloat16)
    for _ in range(5):
        ref = g.generate(Step(0.45, 0.5))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        g.generate(Step(0.45, 0.5))
    ev[1].record()
    torch.cuda.synchronize()
    print(f"Colored B={B} {unit}: {ev[0].elapsed_time(ev[1]) / 20 * 1e3:8.1f} us/draw  group_mb={os.environ.get('SKR_COLORED_GROUP_MB', 'default')}  checksum {ref.float().abs().sum().item():.6e}")
