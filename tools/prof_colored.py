"""Colored noise at the cfg3 shape (256 x 16x128x128 bf16) and the 2-D cfg2 shape -- for rocprofv3 (kernel trace / PMC)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
seeds = list(range(256))
g3 = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, (16, 128, 128), seeds, props=PN.ColoredProps(), dtype=torch.bfloat16)
g2 = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, (1, 128, 128), seeds, props=PN.ColoredProps(), dtype=torch.bfloat16)
for _ in range(6):
    g3.generate(Step(0.45, 0.5)); g2.generate(Step(0.45, 0.5))
torch.cuda.synchronize()
