"""Kernel trace target: Colored draws of two small awkward units (own transforms), 20 draws each after warm-up.
usage: rocprofv3 --kernel-trace --stats -- python3 tools/prof_fft_small.py [unit as a,b,c] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd import _hip
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step

_hip.load()
unit = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (4, 30, 90)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(B)), props=PN.ColoredProps(), dtype=torch.float32)
for _ in range(23):
    out = g.generate(Step(0.45, 0.5))
torch.cuda.synchronize()
print(unit, B, float(out.double().std()))
