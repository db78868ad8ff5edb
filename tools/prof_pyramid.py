"""Pyramid noise at the cfg5 shard shape (64 x 4x256x256 bf16) and the cfg3 shape -- for rocprofv3 (kernel trace / PMC)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
gens = {
    "cfg5 64x(4,256,256)": PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, (4, 256, 256), list(range(64)), props=PN.PyramidProps(), dtype=torch.bfloat16),
    "cfg3 256x(16,128,128)": PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, (16, 128, 128), list(range(256)), props=PN.PyramidProps(), dtype=torch.bfloat16),
}
for name, g in gens.items():
    for _ in range(3): g.generate(Step(0.45, 0.5))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.generate(Step(0.45, 0.5))
    torch.cuda.synchronize()
    print(f"Pyramid {name:24s} {(time.perf_counter() - t0) / 10 * 1e6:8.1f} us/call  (SKR_PYR_MODE={os.environ.get('SKR_PYR_MODE', 'default')})")
