"""one awkward Colored shape on the library's own transforms -- for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
unit = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,66,130").split(","))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(B)), props=PN.ColoredProps(), dtype=torch.float32)
for _ in range(8):
    g.generate(Step(0.45, 0.5))
torch.cuda.synchronize()
