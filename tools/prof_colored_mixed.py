"""Colored noise at 256 x (16, 96, 96) and 64 x (4, 160, 160) bf16 (the mixed-radix plane kernels, compile-time geometry) -- for rocprofv3 (kernel trace / PMC)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
ga = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, (16, 96, 96), list(range(256)), props=PN.ColoredProps(), dtype=torch.bfloat16)
gb = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, (4, 160, 160), list(range(64)), props=PN.ColoredProps(), dtype=torch.bfloat16)
gc = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, (4, 112, 144), list(range(64)), props=PN.ColoredProps(), dtype=torch.bfloat16)
for _ in range(6):
    ga.generate(Step(0.45, 0.5)); gb.generate(Step(0.45, 0.5)); gc.generate(Step(0.45, 0.5))
torch.cuda.synchronize()
