"""Build-container check (needs /root/reference): draw N random wrapper configurations from tools/make_golden.py's sweep grammar,
step each through the imported reference and through skrample_amd on host tensors with the same inputs (teacher-forced), and report
every difference -- results beyond the parity bar, different timesteps, or one side raising where the other does not.
tests/golden/steps_sweep.npz holds the first 64 accepted cases of the same generator; this runs as many as asked.

    python tools/sweep_vs_reference.py [count] [first_seed]
"""

import os
import random
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
warnings.filterwarnings("ignore")

import make_golden as MG  # noqa: E402  (installs the reference loader)

import skrample_amd.diffusers as PD  # noqa: E402
import skrample_amd.scheduling as PS  # noqa: E402
from skrample_amd.sampling import lazy  # noqa: E402
from skrample_amd.sampling import models as PM  # noqa: E402
from skrample_amd.sampling import structured as PT  # noqa: E402

def bf16_ulp(ref: torch.Tensor) -> torch.Tensor:
    mag = ref.float().abs().clamp_min(2.0**-126)
    return torch.exp2(torch.floor(torch.log2(mag)) - 7)


REF = {"W": MG.RD, "T": MG.structured, "S": MG.RS, "M": MG.models, "torch": torch}
OWN = {"W": PD, "T": PT, "S": PS, "M": PM, "torch": torch}


class _Replay:
    def __init__(self, draws):
        self.draws = list(draws)

    def generate(self, step):
        return self.draws.pop(0)

    generate_lazy = generate


def one(seed: int) -> str | None:
    text, dtype, shape, steps_n = MG._sweep_spec(random.Random(seed))
    dt = getattr(torch, dtype)
    g = torch.Generator().manual_seed(seed)
    try:
        r = eval(text, REF)
        r.set_timesteps(steps_n)
        r_times = r.timesteps.clone()
        r_err = None
    except Exception as err:  # noqa: BLE001
        r_err = err
    try:
        p = eval(text, OWN)
        p.set_timesteps(steps_n)
        p_times = p.timesteps.clone()
        p_err = None
    except Exception as err:  # noqa: BLE001
        p_err = err
    if r_err or p_err:
        if type(r_err) is not type(p_err):
            return f"set_timesteps: reference {r_err!r}, here {p_err!r}"
        return None
    if r_times.shape != p_times.shape or (r_times - p_times).abs().max() > 1e-9:
        return f"timesteps differ: {r_times.tolist()} vs {p_times.tolist()}"
    x = torch.randn(shape, generator=g).to(dt)
    n = len(r_times)
    outs = [torch.randn(shape, generator=g).to(dt) for _ in range(n)]
    noises = [torch.randn(shape, generator=g) for _ in range(n)]
    r._noise_generator, p._noise_generator = MG._Injected(noises), _Replay(noises)
    for i, (tr, tp) in enumerate(zip(r_times, p_times)):
        try:
            ref = r.step(outs[i], tr, x, return_dict=False)
            r_err = None
        except Exception as err:  # noqa: BLE001
            r_err = err
        try:
            got = p.step(outs[i], tp, x, return_dict=False)
            got = [torch.as_tensor(v.materialize() if isinstance(v, lazy.LazyTensor) else v) for v in got]
            p_err = None
        except Exception as err:  # noqa: BLE001
            p_err = err
        if r_err or p_err:
            if r_err and p_err:
                return None
            finite = r_err is None and all(torch.isfinite(v.float()).all() for v in ref)
            if p_err and not finite:
                return None  # (singular point: the reference hands back inf / nan, the engine raises -- INTEGRATION.md)
            return f"step {i}: reference {r_err!r}, here {p_err!r}"
        if not all(torch.isfinite(v.float()).all() for v in ref):
            return None
        for name, a, b in zip(("prev", "pred"), got, ref):
            if a.dtype != b.dtype or a.shape != b.shape:
                return f"step {i} {name}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}"
            a64, b64 = a.double(), b.double()
            scale = b64.abs().max().clamp_min(1e-30)
            if dt in (torch.float32, torch.float64):
                allowed = 1e-5 * scale  # north_star's bar: relative inf-norm error
            else:  # one unit in the last place of the reference's 16-bit result (+ 1e-6 max|ref| for cancellation residues), as tests/test_step_gpu.py
                allowed = bf16_ulp(b).double() * (1 if dt == torch.bfloat16 else 2.0**-3) + 1e-6 * scale
            bad = ((a64 - b64).abs() > allowed).sum().item()
            if bad:
                return f"step {i} {name}: {bad} elements beyond the bar (max diff {(a64 - b64).abs().max().item():.3g}, scale {scale.item():.3g})"
        x = ref[0]
    return None


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    found = 0
    for seed in range(first, first + count):
        text = MG._sweep_spec(random.Random(seed))
        try:
            why = one(seed)
        except Exception as err:  # noqa: BLE001
            why = f"harness error {err!r}"
        if why:
            found += 1
            print(f"seed {seed}: {text}\n    {why}", flush=True)
    print(f"{count} configurations, {found} differences")
