"""GPU box: randomised shapes / properties for the Offset, Pyramid, Colored and Brownian generators against the oracle
(usage: python tests/soak_noise.py [n_cases]); the fixed cases live in tests/test_noise_gpu.py."""
import os, random, sys, traceback
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import test_noise_gpu as T
from test_noise_gpu import PN, ON, Step, spec_normal, rel
from skrample_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(os.environ.get("SOAK_SEED", "2026")))
bad = 0

def case_offset():
    nd = rng.randint(1, 4)
    unit = tuple(rng.choice((1, 2, 3, 4, 5, 8, 16, 24)) for _ in range(nd))
    dims = tuple(sorted(rng.sample(range(nd), rng.randint(1, nd))))
    props = PN.OffsetProps(dims=dims, strength=rng.choice((0.2, 0.7, 1.5)))
    seeds = [rng.randrange(2**63) for _ in range(rng.randint(1, 3))]
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Offset, unit, seeds, props=props, dtype=torch.float32)
    for n in range(2):
        got = g.generate(None).cpu()
        refs = [ON.offset_noise(unit, ON.Replay([spec_normal(s, n * 256 + 1, ON.offset_shape(unit, dims)), spec_normal(s, n * 256, unit)]).randn, dims, props.strength) for s in seeds]
        assert rel(got, torch.stack(refs)) < 1e-5, ("offset", unit, props, n, rel(got, torch.stack(refs)))

def case_pyramid():
    lead = rng.choice((1, 2, 4))
    h, w = rng.choice((8, 16, 24, 40, 64, 100, 128, 9, 30, 45)), rng.choice((8, 16, 24, 40, 64, 100, 128, 10, 18, 90, 33))
    unit = (lead, h, w)
    kw = dict(strength=rng.choice((0.3, 0.6, 0.9)), depth=rng.choice((99, 1, 2)))
    (T.test_pyramid if w % 4 == 0 else T.test_pyramid_any_shape)(unit, kw, dev)

def case_colored():
    nd = rng.choice((2, 3, 3, 4))
    pow2 = rng.random() < 0.7
    pick = (lambda: rng.choice((4, 8, 16, 32, 64, 128))) if pow2 else (lambda: rng.choice((6, 12, 20, 24, 40, 48, 96)))
    unit = tuple(pick() for _ in range(nd))
    if nd == 3:
        unit = (rng.choice((1, 2, 4, 8, 16, 32)) if pow2 else rng.choice((1, 3, 4, 6)),) + unit[1:]
    if nd == 4:
        unit = (rng.choice((2, 3, 4, 16)), rng.choice((2, 5, 8))) + tuple(min(d, 24) for d in unit[2:])
    T.test_colored.__wrapped__(unit, dev) if hasattr(T.test_colored, "__wrapped__") else T.test_colored(unit, dev)

def case_colored_extreme():
    "power-of-two units of extreme aspect (one-quad rows, one row pair, long lines, deep outer axes): every route of skr_noise_colored"
    nd = rng.choice((2, 3, 3))
    total = 1
    unit = []
    for ax in range(nd):
        hi = 12 if ax == nd - 1 else (7 if nd == 3 and ax == 0 else 12)
        lo = 2 if ax == nd - 1 else (1 if ax == nd - 2 else 0)
        e = rng.choice((lo, lo, lo + 1, rng.randint(lo, hi), hi if rng.random() < 0.15 else rng.randint(lo, min(hi, 8))))
        unit.append(1 << e)
    while np.prod(unit) > (1 << 21):
        i = int(np.argmax(unit))
        unit[i] //= 2
    T.test_colored.__wrapped__(tuple(unit), dev) if hasattr(T.test_colored, "__wrapped__") else T.test_colored(tuple(unit), dev)

def case_pyramid_extreme():
    "narrow, short and near-limit planes for the LDS kernels and the any-shape kernels"
    lead = rng.choice((1, 2, 3))
    h = rng.choice((1, 2, 3, 4, 5, 7, 8, 16, 31, 64, 200, 380))
    w = rng.choice((4, 8, 12, 20, 28, 36, 252, 380, 2, 3, 5, 6, 7, 9, 13, 250))
    if h * w > 380 * 380 or h * w < 2:
        h = 8
    unit = (lead, h, w)
    kw = dict(strength=rng.choice((0.3, 0.6, 0.9)), depth=rng.choice((99, 1, 2, 5)))
    seeds = [rng.randrange(2**63) for _ in range(2)]
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=PN.PyramidProps(**kw), dtype=torch.float32)
    for n in range(2):
        ref = torch.stack([T.pyramid_reference(unit, s, n * 256, **kw) for s in seeds])
        got = g.generate(None).cpu()
        assert rel(got, ref) < 2e-5, ("pyramid", unit, kw, n, rel(got, ref))

def case_colored_mixed():
    "even heights, widths 4 * k: the mixed-radix plane kernel where the odd part is <= 63 and the plane fits LDS, the own N-D transform otherwise"
    while True:
        h, w = 2 * rng.randint(3, 100), 4 * rng.randint(1, 50)
        if h * w <= 36000:
            break
    lead = rng.choice((None, 1, 2, 4, 8))
    unit = (h, w) if lead is None else (lead, h, w)
    T.test_colored.__wrapped__(unit, dev) if hasattr(T.test_colored, "__wrapped__") else T.test_colored(unit, dev)

def case_colored_awkward():
    "any side lengths -- odd, prime, next to a power of two, not multiples of 4: the library's own any-length transforms (skr_fft_own.hip)"
    nd = rng.choice((1, 2, 2, 3, 3, 4))
    pick = lambda hi: rng.choice((rng.randint(2, hi), rng.randint(2, hi), rng.choice((3, 5, 7, 11, 13, 17, 31, 33, 63, 65, 67, 97, 127, 129, 131, 255, 257)), 2 * rng.randint(1, hi // 2) + 1))
    while True:
        unit = tuple(min(pick(300), 300) for _ in range(nd))
        if nd == 1:
            unit = (rng.choice((1, 2, 3)), rng.choice((rng.randint(2, 2048), 1025, 2047, 2048, 1031)))
        if nd == 4:
            unit = (rng.choice((2, 3, 4)), rng.randint(2, 9)) + tuple(min(d, 40) for d in unit[2:])
        if int(np.prod(unit)) <= (1 << 19):
            break
    T.test_colored.__wrapped__(unit, dev) if hasattr(T.test_colored, "__wrapped__") else T.test_colored(unit, dev)

def diag_colored(unit):
    "which side moved?  device result vs the oracle (torch CPU FFT) vs an independent float64 numpy evaluation of the same pipeline"
    if unit is None:
        return
    seeds = [31, 32]
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
    got = g.generate(None).cpu().double()
    whites = [spec_normal(s, 0, unit) for s in seeds]
    ref = torch.stack([ON.colored_noise(unit, lambda shape, w=w: w, None) for w in whites]).double()
    outs = []
    for w in whites:
        x = w.double().numpy().squeeze()
        spec = np.fft.rfftn(x)
        axes = [np.abs(np.fft.fftfreq(d)) if i < x.ndim - 1 else np.arange(d // 2 + 1) / d for i, d in enumerate(x.shape)]
        rad = np.sqrt(sum(a ** 2 for a in np.meshgrid(*axes, indexing="ij")))
        rad = rad / rad.max()
        clip = 0.5 / max(sum(x.shape) / x.ndim, 4.0)
        col = np.fft.irfftn(spec * np.clip(rad, clip, None) ** (-0.25 / 2.0), s=x.shape)
        col *= x.std(ddof=1) / col.std(ddof=1)
        outs.append(torch.from_numpy(col.reshape(w.shape)))
    ref2 = torch.stack(outs)
    print("   diag", unit, "device vs oracle", rel(got, ref), " device vs numpy64", rel(got, ref2), " oracle vs numpy64", rel(ref, ref2))
    # the same through colorize_noise (one sample per call, its own workspaces)
    c = torch.stack([PN.Colored.colorize_noise(w.to(dev), exponent=0.25).cpu().double() for w in whites])
    print("   diag colorize_noise vs numpy64", rel(c, ref2))
    # per (sample, outermost index) error of the device result, and torch's own GPU FFT (also rocFFT) of the same batched inner shape
    if len(unit) == 4:
        err = (got - ref2).abs().reshape(len(seeds), unit[0], -1).max(dim=2).values / ref2.abs().max()
        print("   diag error by (sample, outer index):", [[float(f"{v:.1e}") for v in row] for row in err.tolist()])
        xw = torch.stack(whites).reshape(-1, *unit[1:]).float()
        gpu = torch.fft.rfftn(xw.to(dev), dim=(-3, -2, -1)).cpu()
        cpu = torch.fft.rfftn(xw, dim=(-3, -2, -1))
        print("   diag torch.fft.rfftn on the GPU vs CPU:", (gpu - cpu).abs().max().item() / cpu.abs().max().item())


def case_brownian():
    unit = tuple(rng.choice((1, 2, 3, 4, 8, 16)) for _ in range(rng.randint(1, 3)))
    seeds = [rng.randrange(2**63) for _ in range(rng.randint(1, 3))]
    ms = rng.choice((100, 1000, 10_000))
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, props=PN.BrownianProps(max_steps=ms), dtype=torch.float32)
    n = rng.randint(3, 40)
    k = rng.randrange(0, n - 2)
    for st in (Step.from_int(k, n), Step.from_int(k + 1, n), Step(rng.random(), rng.random()), Step.from_int(k + 1, n)):
        if abs(st.time_to - st.time_from) < 2.0 / ms:
            continue
        got = g.generate(st).cpu().double()
        ref = torch.stack([ON.brownian_noise(s, unit, st, ms, grid=g._state["brownian_grid"]) for s in seeds])  # (the first query fixed the path: partition or dyadic)
        # unit-variance output = scale * (W(to) - W(from)) with both path values accumulated in fp32: the error floor is absolute
        # (~1e-6), so on units of a few elements, whose largest value can be small by chance, it is measured against >= 1
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
        assert err < 1e-5, ("brownian", unit, st, ms, err)

for i in range(n_cases):
    if i % 20 == 0:
        print(f"case {i} of {n_cases}, failures so far: {bad}", flush=True)  # (a silent GPU box is taken for a hung one)
    for fn in ((case_offset, case_pyramid, case_pyramid_extreme, case_colored, case_colored_mixed, case_colored_extreme, case_colored_awkward, case_brownian) if not os.environ.get("SOAK_ONLY") else (globals()["case_" + os.environ["SOAK_ONLY"]],)):
        state = rng.getstate()
        try:
            fn()
        except Exception as e:
            bad += 1
            print("FAIL", fn.__name__, i, type(e).__name__, str(e)[:400])
            after = rng.getstate()
            rng.setstate(state)  # the same case again, in the same process: a persistent or a transient failure?
            try:
                fn()
                print("   the same case again: passes")
            except Exception as e2:
                print("   the same case again: fails again --", str(e2)[:200])
            rng.setstate(after)
            if fn in (case_colored, case_colored_mixed, case_colored_extreme, case_colored_awkward):
                diag_colored(e.args[0][0] if e.args and isinstance(e.args[0], tuple) else None)
lib = _hip.load()
print("done, failures:", bad, "| hipFFT plans made:", lib.skr_stat(b"hipfft_plans"), "hipFFT transforms run:", lib.skr_stat(b"hipfft_execs"), "own N-D transforms run:", lib.skr_stat(b"own_fft_execs"))
assert lib.skr_stat(b"hipfft_plans") == 0 and lib.skr_stat(b"hipfft_execs") == 0  # the vendor FFT runs only when asked for (skr_set_tuning "hipfft" 1)
