"""Generator timings on the MI355X (not the headline metric): per-call time and GB/s of output."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
dev = torch.device('cuda:0')
def timeit(name, gen, step, n=20):
    steps = step if isinstance(step, list) else [step] * (n + 3)
    for i in range(3): gen.generate(steps[i])
    torch.cuda.synchronize(); t=time.perf_counter()
    for i in range(n): out = gen.generate(steps[3 + i])
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/n
    print(f"{name:40s} {dt*1e3:8.3f} ms/call  {out.numel()/dt/1e9:7.2f} Gelem/s  out {tuple(out.shape)} {out.dtype}")
for B, unit in [(256,(16,128,128)), (64,(4,128,128)), (64,(4,256,256)), (256,(4,128,128))]:
    seeds = list(range(B))
    timeit(f"Random   B={B} {unit}", PN.BatchTensorNoise.from_batch_inputs(PN.Random, unit, seeds, dtype=torch.bfloat16), None)
    timeit(f"Offset   B={B} {unit}", PN.BatchTensorNoise.from_batch_inputs(PN.Offset, unit, seeds, props=PN.OffsetProps(), dtype=torch.bfloat16), None)
    timeit(f"Pyramid  B={B} {unit}", PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=PN.PyramidProps(), dtype=torch.bfloat16), None)
    timeit(f"Colored  B={B} {unit}", PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.bfloat16), Step(0.45,0.5))
    timeit(f"Brownian B={B} {unit}", PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, props=PN.BrownianProps(), dtype=torch.bfloat16), Step(0.35,0.4))  # (not 0.5: a dyadic endpoint has a 2-node path)
    timeit(f"Brownian B={B} {unit} sequential", PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, props=PN.BrownianProps(), dtype=torch.bfloat16), [Step.from_int(k, 30) for k in range(2, 28)])
