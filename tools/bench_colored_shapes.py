import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from skrample_amd import _hip
if os.environ.get("SKR_LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["SKR_LIB"])
from skrample_amd.pytorch import noise as PN
from skrample_amd.common import Step
for batch, unit in ((64, (4, 96, 96)), (64, (4, 128, 128)), (256, (16, 96, 96)), (256, (16, 128, 128)), (64, (4, 160, 160)), (64, (4, 192, 192)), (64, (4, 256, 256)),
                    (64, (4, 112, 144)), (64, (4, 104, 152)), (64, (4, 96, 168)), (64, (4, 80, 192)), (64, (4, 120, 120)), (64, (4, 124, 116)), (64, (4, 136, 184)), (64, (4, 148, 172)), (64, (4, 244, 68))):
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, list(range(batch)), props=PN.ColoredProps(), dtype=torch.bfloat16)
    st = Step(0.45, 0.5)
    for _ in range(3): g.generate(st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.generate(st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    n = batch
    for d in unit: n *= d
    print(f"Colored B={batch} {unit}: {dt*1e3:.3f} ms  {n/dt/1e9:.1f} Gelem/s", flush=True)
