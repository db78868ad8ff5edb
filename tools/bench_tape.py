"""The sampler-level API on bf16 device tensors: the op-tape launch (native.mode = "auto": the reference's rounded ops, skr_tape_launch) against the
fused form (native.mode = "never": skr_step_launch) of the same steps, 256 x 4 x 128 x 128 -- time per step and HBM rate on the bytes each moves.
usage: python tools/bench_tape.py   (one GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skrample_amd.scheduling as PS
from skrample_amd import _hip
from skrample_amd.common import Step
from skrample_amd.sampling import models as PM, native
from skrample_amd.sampling import structured as PT

_hip.load()
dev = torch.device("cuda:0")
shape = (256, 4, 128, 128)
numel = 256 * 4 * 128 * 128
g = torch.Generator(device=dev).manual_seed(0)
bufs = [[torch.randn(shape, device=dev, generator=g).bfloat16() for _ in range(3)] for _ in range(8)]  # (x, out, noise) x 8 rotating sets: 8 x 100 MB
cases = [("Euler", PT.Euler(), PM.NoiseModel(), PS.Scaled(), 0), ("DPM-2 SDE", PT.DPM(order=2, stochasticity=1), PM.NoiseModel(), PS.Karras(PS.Scaled()), 1),
         ("Adams-4", PT.Adams(order=4), PM.VelocityModel(), PS.ZSNR(), 3), ("UniPC-3 SDE", PT.UniPC(order=3, stochasticity=1), PM.FlowModel(), PS.Linear(), 2)]
print(f"{'sampler':14s} {'mode':6s} {'us/step':>9s} {'operands':>9s} {'GB/s':>8s}")
for name, sampler, model, schedule, hist in cases:
    for mode in ("auto", "never"):
        native.mode = mode
        steps = 20
        prev = []
        for i in range(hist + 1):  # history of the steady state
            x, out, nz = bufs[i % 8]
            prev.append(sampler.sample(x, out, Step.from_int(i, steps), model, schedule, nz if sampler.require_noise else None, tuple(prev)))
        prev = prev[-max(hist, 1):] if hist else []
        i0 = hist + 1
        def run(n):
            before = native.launches
            for k in range(n):
                x, out, nz = bufs[(i0 + k) % 8]
                r = sampler.sample(x, out, Step.from_int(min(i0 + 2, steps - 2), steps), model, schedule, nz if sampler.require_noise else None, tuple(prev))
                torch.as_tensor(r.final)
            return native.launches - before
        run(17)  # every buffer set twice: a set whose tensors coincide with the history's changes the operand count, and the first launch of a kernel variant costs ~20 ms
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        taped = run(40)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / 40 * 1e3
        n_in = 2 + (1 if sampler.require_noise else 0) + (0 if not hist else len(prev) * (2 if "UniPC" in name else 1))
        n_out = 3 if "UniPC" in name else 1
        gbs = (n_in + n_out) * numel * 2 / (us * 1e-6) / 1e9
        print(f"{name:14s} {mode:6s} {us:9.1f} {n_in:>4d}+{n_out:<4d} {gbs:8.0f}   (tape launches: {taped})", flush=True)
native.mode = "auto"
