"""GPU box: the device path against the package's own host executor over the sweep grammar (tests/sweep_grammar.py) -- usage:
python tests/soak_sweep.py [first_seed last_seed [native | reuse | plain [grow]]].  The host side of every case is what tools/sweep_vs_reference.py compares with the imported
reference in the build container (1500 configurations, 0 differences), so device == host here carries the reference's answer to seeds
the fixture (tests/golden/steps_sweep.npz, 64 cases) does not hold.  Teacher-forced: each step sees the host run's inputs."""
import os
import random
import sys
import traceback

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_step_gpu as T  # noqa: E402
from sweep_grammar import native_spec, sweep_spec  # noqa: E402

from skrample_amd import _hip  # noqa: E402
from skrample_amd.sampling import lazy  # noqa: E402

_hip.load()
dev = torch.device("cuda:0")


def settle(v):
    return torch.as_tensor(v.materialize() if isinstance(v, lazy.LazyTensor) else v)


NATIVE = len(sys.argv) > 3 and sys.argv[3] == "native"  # compute_scale=None on 16-bit tensors: the tape on the device against torch's own ops on the host, bit for bit
REUSE = len(sys.argv) > 3 and sys.argv[3] == "reuse"  # one scheduler object over three runs: from a random later index, from the start, from the start with float timesteps


GROW = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # planes GROW x GROW times larger: the launches of large tensors (several trips per lane, two words per lane on the tape)


def one(seed: int) -> str:
    text, dtype, shape, steps_n = (native_spec if NATIVE else sweep_spec)(random.Random(seed))
    shape = (*shape[:-2], shape[-2] * GROW, shape[-1] * GROW + (seed % 3 if GROW > 1 else 0))  # (ragged rows too)
    dt = getattr(torch, dtype)
    g = torch.Generator().manual_seed(seed)
    try:
        host, card = eval(text, T.SWEEP_NAMES), eval(text, T.SWEEP_NAMES)
        host.set_timesteps(steps_n)
        card.set_timesteps(steps_n)
        times = host.timesteps
    except (ZeroDivisionError, ValueError, AssertionError, IndexError, AttributeError, TypeError):
        return "refused"
    if not torch.isfinite(times).all():
        return "non-finite"  # (a Karras / Exponential ramp over one step: nan timesteps, in the reference too)
    assert torch.equal(times, card.timesteps.cpu())
    n = len(times)
    if REUSE:
        # what the replayed steps (step programs, _fast_step) remember must hold for the next run of the same object whatever that run looks like
        first = 0 if "RK" in text else random.Random(seed ^ 0x5eed).randrange(n)  # (the Runge-Kutta wrappers insist on the schedule's order, as the reference's do)
        for lap, (start, as_float) in enumerate(((first, False), (0, False), (0, True))):
            host.set_timesteps(steps_n)
            card.set_timesteps(steps_n)
            x = torch.randn(shape, generator=g).to(dt)
            outs = [torch.randn(shape, generator=g).to(dt) for _ in range(n)]
            noises = [torch.randn(shape, generator=g) for _ in range(n)]
            host._noise_generator, card._noise_generator = T.Injected(noises[start:], "cpu"), T.Injected(noises[start:], dev)
            for i in range(start, n):
                t = float(times[i]) if as_float else times[i]
                try:
                    ref = [settle(v) for v in host.step(outs[i], t, x, return_dict=False)]
                except (ZeroDivisionError, np.linalg.LinAlgError):
                    try:
                        card.step(outs[i].to(dev), t, x.to(dev), return_dict=False)
                    except (ZeroDivisionError, np.linalg.LinAlgError):
                        return "singular"
                    raise AssertionError(f"lap {lap} step {i}: the host path refuses a singular point, the device path does not")
                got = [settle(v) for v in card.step(outs[i].to(dev), t, x.to(dev), return_dict=False)]
                if not all(torch.isfinite(v.float()).all() for v in ref):
                    return "non-finite"
                for name, a, b in zip(("prev_sample", "pred_original_sample"), got, ref):
                    T.assert_close(a, b, dt, f"lap {lap} (from {start}{', float timesteps' if as_float else ''}) step {i} {name}", flips=0.2)
                x = ref[0]
        return "ok"
    x = torch.randn(shape, generator=g).to(dt)
    outs = [torch.randn(shape, generator=g).to(dt) for _ in range(n)]
    noises = [torch.randn(shape, generator=g).to(dt if NATIVE else torch.float32) for _ in range(n)]
    host._noise_generator, card._noise_generator = T.Injected(noises, "cpu"), T.Injected(noises, dev)
    # SPC's signed-power blend, sign(s)|s|^p wp + sign(c)|c|^p wc to the power 1/p, has no bound on its conditioning: where the two terms nearly cancel the
    # root multiplies the rounding of the powers (libm's on the host, the device's own here) without limit, and a large tensor always holds such elements
    # (seed 1300173 at 1.5 M elements: 2.4e-5 of max|ref| between device and host in fp32).  There the yardstick is a float64 host run: the device must be
    # as close to it as the fp32 host run is
    exact = None
    if not NATIVE and "T.SPC(" in text and "power=1," not in text and dt == torch.float32 and "compute_scale=torch.float64" not in text:
        exact = eval(text, T.SWEEP_NAMES)
        exact.compute_scale = torch.float64
        exact.set_timesteps(steps_n)
        exact._noise_generator = T.Injected(noises, "cpu")
    for i, t in enumerate(times):
        try:
            ref = [settle(v) for v in host.step(outs[i], t, x, return_dict=False)]
        except (ZeroDivisionError, np.linalg.LinAlgError):
            try:
                card.step(outs[i].to(dev), t, x.to(dev), return_dict=False)
            except (ZeroDivisionError, np.linalg.LinAlgError):
                return "singular"
            raise AssertionError(f"step {i}: the host path refuses a singular point, the device path does not")
        got = [settle(v) for v in card.step(outs[i].to(dev), t, x.to(dev), return_dict=False)]
        if not all(torch.isfinite(v.float()).all() for v in ref):
            return "non-finite"
        for name, a, b in zip(("prev_sample", "pred_original_sample"), got, ref):
            assert a.is_cuda, name
            if NATIVE:
                assert a.dtype == b.dtype and torch.equal(a.cpu(), b), (f"step {i} {name}", int((a.cpu() != b).sum()))
            elif exact is None:
                T.assert_close(a, b, dt, f"step {i} {name}", flips=0.2)
        if exact is not None:
            try:
                wide = [settle(v).double() for v in exact.step(outs[i], t, x, return_dict=False)]
            except (ZeroDivisionError, np.linalg.LinAlgError):
                return "singular"
            for name, a, b, c in zip(("prev_sample", "pred_original_sample"), got, ref, wide):
                scale = c.abs().max().clamp_min(1e-30)
                ours, theirs = ((a.cpu().double() - c).abs().max() / scale).item(), ((b.double() - c).abs().max() / scale).item()
                assert a.dtype == dt and ours <= 2 * theirs + 1e-5, (  # (the parity bar itself on top of the host run's own distance: seed 5108667, the power-2 blend of a flow-derivative predictor where its terms cancel, is 2.3e-6 on the device -- exp2 / log2 powers -- beside the host's 2.7e-7)
                    f"step {i} {name}: {ours:.3g} from the float64 host run, the fp32 host run {theirs:.3g}")
        x = ref[0]
    return "ok"


if __name__ == "__main__":
    first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800000, 800400)
    tally: dict = {}
    bad = 0
    for seed in range(first, last):
        try:
            kind = one(seed)
        except Exception:  # noqa: BLE001
            bad += 1
            kind = "FAILED"
            print(f"seed {seed}: {sweep_spec(random.Random(seed))}", flush=True)
            traceback.print_exc(limit=3)
        tally[kind] = tally.get(kind, 0) + 1
        if (seed - first) % 100 == 99:
            print(f"... {seed - first + 1} cases: {tally}", flush=True)
    torch.cuda.synchronize()
    print(f"sweep soak over seeds {first}..{last - 1}: {tally}; {bad} failures")
    sys.exit(1 if bad else 0)
