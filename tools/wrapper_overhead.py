"""Host cost of SkrampleWrapperScheduler.step() on the workload the reference's own timing harness uses (reference
scripts/overhead.py:12-21): an Euler sampler over FlowShift(Beta(ZSNR())), a 1000-point schedule, one-element tensors.

    python tools/wrapper_overhead.py                 this package, host-resident tensors (its host executor; no GPU needed)
    python tools/wrapper_overhead.py --device cuda   this package, device tensors (one fused launch per step)
    python tools/wrapper_overhead.py --reference     the reference's classes on the same workload (build container only:
                                                     /root/reference is imported through tools/ref_loader.py)

One line per configuration: microseconds per step() of each repetition, and the best."""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import torch  # noqa: E402

cli = argparse.ArgumentParser()
cli.add_argument("--device", default="cpu")
cli.add_argument("--reference", action="store_true")
cli.add_argument("--points", type=int, default=1000, help="schedule length")
cli.add_argument("--repeats", type=int, default=5)
opt = cli.parse_args()

if opt.reference:
    import ref_loader

    ref_loader.install()
    import skrample.diffusers as wrappers
    import skrample.scheduling as schedules
    from skrample.sampling import structured as samplers

    who = "reference"
else:
    import skrample_amd.diffusers as wrappers
    import skrample_amd.scheduling as schedules
    from skrample_amd.sampling import structured as samplers

    who = f"skrample_amd on {opt.device}"
target = torch.device(opt.device)


def one_pass() -> float:
    "seconds per step over one full schedule (tensor creation inside the loop, as in the reference's harness)"
    scheduler = wrappers.SkrampleWrapperScheduler(samplers.Euler(), schedules.FlowShift(schedules.Beta(schedules.ZSNR())))
    scheduler.set_timesteps(opt.points)
    marks = scheduler.timesteps
    began = time.perf_counter()
    for position in range(len(marks)):
        latent = torch.rand([1], device=target)
        predicted = torch.rand([1], device=target)
        scheduler.step(predicted, marks[position], latent, return_dict=False)
    if target.type == "cuda":
        torch.cuda.synchronize()
    return (time.perf_counter() - began) / len(marks)


costs = [one_pass() * 1e6 for _ in range(opt.repeats)]
print(f"{who:28s} us/step: " + "  ".join(f"{c:8.1f}" for c in costs) + f"   best {min(costs):.1f}")
