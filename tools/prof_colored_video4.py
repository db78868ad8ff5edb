import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
for batch, unit in ((4, (16, 21, 90, 160)), (8, (16, 13, 60, 104)), (8, (16, 21, 64, 64))):
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(batch)), props=PN.ColoredProps(), dtype=torch.bfloat16)
    for _ in range(6): g.generate(Step(0.45, 0.5))
torch.cuda.synchronize()
