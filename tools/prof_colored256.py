import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, (4, 256, 256), list(range(64)), props=PN.ColoredProps(), dtype=torch.bfloat16)
for _ in range(6): g.generate(Step(0.45, 0.5))
torch.cuda.synchronize()
