"""GPU box: a serving-length run (usage: python tests/soak_leak.py [runs]) -- many sampling runs back to back through every wrapper kind
(new scheduler object per run or one reused, device-resident and host timesteps, every noise type, a captured loop replayed, the sampler-level
API on 16-bit tensors), watching what must stay flat: device memory held by live tensors, the allocator's reservation, the host RSS, the
library's plan / workspace counters.  Prints one line per checkpoint; exits 1 when the peaks of the last third of the run exceed those of the third before."""
import os
import sys

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root]
import psutil  # noqa: E402
import torch  # noqa: E402

import skrample_amd.diffusers as PD  # noqa: E402
import skrample_amd.scheduling as PS  # noqa: E402
from skrample_amd import _hip  # noqa: E402
from skrample_amd.common import Step  # noqa: E402
from skrample_amd.graphs import CapturedLoops  # noqa: E402
from skrample_amd.pytorch import noise as PN  # noqa: E402
from skrample_amd.sampling import models as PM  # noqa: E402
from skrample_amd.sampling import structured as PT  # noqa: E402

_hip.load()
dev = torch.device("cuda:0")
shape = (8, 4, 64, 64)
g = torch.Generator().manual_seed(1)
x0 = torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)
outs = [torch.randn(shape, generator=g).to(torch.bfloat16).to(dev) for _ in range(8)]
net = lambda x, t: x * 0.75 - (t / 2000) * x.abs()  # noqa: E731

MAKERS = [
    lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())),
    lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps()),
    lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()),
    lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=4, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps()),
    lambda: PD.SkrampleWrapperScheduler(PT.Euler(stochasticity=1), PS.Scaled(), noise_type=PN.Offset, noise_props=PN.OffsetProps()),
    lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Brownian, noise_props=PN.BrownianProps()),
    lambda: PD.SkrampleWrapperScheduler(PT.SPC(), PS.Scaled()),
]
reused = [mk() for mk in MAKERS]
loops = CapturedLoops(MAKERS[0], net, x0, seeds=list(range(shape[0])), keep=3)


def one_run(r: int) -> None:
    k = r % len(MAKERS)
    w = MAKERS[k]() if r % 2 else reused[k]
    steps = 5 + r % 7
    w.set_timesteps(steps, device=dev) if r % 3 == 0 else w.set_timesteps(steps)
    x = x0
    seeds = [torch.Generator().manual_seed(r * 100 + b) for b in range(shape[0])]
    for i, t in enumerate(w.timesteps):
        x = torch.as_tensor(w.step(outs[i % 8], t, x, generator=seeds, return_dict=False)[0])
    loops(x0, 4 + r % 5, seeds=list(range(r, r + shape[0])))  # five run lengths over three resident graphs: captures keep happening
    sampler, prev, x = PT.DPM(order=2, stochasticity=1), [], x0  # the sampler-level API on 16-bit tensors (tape launches)
    for i in range(4):
        rec = sampler.sample(x, outs[i], Step.from_int(i, 4), PM.NoiseModel(), PS.Scaled(), outs[7 - i], tuple(prev))
        prev.append(rec)
        x = rec.final


def snapshot() -> dict:
    torch.cuda.synchronize(dev)
    lib = _hip.load()
    stats = {}
    for key in (b"hipfft_plans", b"hipfft_execs"):
        try:
            stats[key.decode()] = int(lib.skr_stat(key))
        except Exception:  # noqa: BLE001
            pass
    return {"allocated_mb": torch.cuda.memory_allocated(dev) / 2**20, "reserved_mb": torch.cuda.memory_reserved(dev) / 2**20, "rss_mb": psutil.Process().memory_info().rss / 2**20, **stats}


if __name__ == "__main__":
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    marks = []
    for r in range(runs):
        one_run(r)
        if r % (runs // 15) == runs // 15 - 1:
            marks.append(snapshot())
            print(f"run {r + 1:5d}: " + "  ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in marks[-1].items()), flush=True)
    # what is live at a checkpoint depends on which wrapper kinds the run counter stopped at (15-22 MB here): compare the PEAKS of the last
    # third of the checkpoints with those of the third before
    n = len(marks)
    mid, last = marks[n // 3 : 2 * n // 3], marks[2 * n // 3 :]
    grow = {k: max(m[k] for m in last) - max(m[k] for m in mid) for k in ("allocated_mb", "reserved_mb", "rss_mb")}
    ok = grow["allocated_mb"] <= 1.0 and grow["reserved_mb"] <= 1.0 and grow["rss_mb"] <= 8.0
    print(f"{runs} runs; peaks of the last third against the third before: " + ", ".join(f"{k} {v:+.1f}" for k, v in grow.items()) + ("  -- flat" if ok else "  -- GROWING"))
    sys.exit(0 if ok else 1)
