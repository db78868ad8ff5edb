"""Wrapper step() wall-clock on the reference's own overhead workload (reference scripts/overhead.py:12-21):
Euler + FlowShift(Beta(ZSNR())), 1000 timesteps, `[1]`-element tensors, perf_counter_ns around the loop.

    python tools/overhead.py                 this package, host-resident tensors (the host executor, no GPU needed)
    python tools/overhead.py --device cuda   this package, device tensors (one fused launch per step)
    python tools/overhead.py --reference     the reference's classes on the same workload (build container only:
                                             /root/reference is imported through tools/ref_loader.py)

Prints ns per step for each of five runs, like the reference's script prints ns per 1000-step loop."""
import argparse, os, sys
from time import perf_counter_ns

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--device", default="cpu")
ap.add_argument("--reference", action="store_true")
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--runs", type=int, default=5)
args = ap.parse_args()

if args.reference:
    import ref_loader
    ref_loader.install()  # import hook for /root/reference
    from skrample.diffusers import SkrampleWrapperScheduler
    from skrample.sampling.structured import Euler
    from skrample.scheduling import ZSNR, Beta, FlowShift
    label = "reference"
else:
    from skrample_amd.diffusers import SkrampleWrapperScheduler
    from skrample_amd.sampling.structured import Euler
    from skrample_amd.scheduling import ZSNR, Beta, FlowShift
    label = f"skrample_amd[{args.device}]"
dev = torch.device(args.device)


def bench_wrapper() -> int:
    wrapper = SkrampleWrapperScheduler(Euler(), FlowShift(Beta(ZSNR())))
    wrapper.set_timesteps(args.steps)
    clock = perf_counter_ns()
    for timestep in wrapper.timesteps:
        output, sample = torch.rand([1], device=dev), torch.rand([1], device=dev)
        wrapper.step(output, timestep, sample, return_dict=False)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return perf_counter_ns() - clock


runs = [bench_wrapper() / args.steps for _ in range(args.runs)]
print(f"{label:24s} ns/step: " + "  ".join(f"{r:9.0f}" for r in runs) + f"   best {min(runs) / 1e3:.1f} us/step")
