"""rocprofv3 target: Colored draws of a video-latent unit (channels x frames x height x width), plane-kernel route or hipFFT (SKR_FFT_NO_PLANES=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
unit = tuple(int(v) for v in os.environ.get("UNIT", "16,21,90,160").split(","))
batch = int(os.environ.get("BATCH", "1"))
g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(batch)), props=PN.ColoredProps(), dtype=torch.bfloat16)
for _ in range(12):
    g.generate(Step(0.45, 0.5))
torch.cuda.synchronize()
