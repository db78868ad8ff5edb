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
rng = random.Random(2026)
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
        ref = torch.stack([ON.brownian_noise(s, unit, st, ms) for s in seeds])
        # unit-variance output = scale * (W(to) - W(from)) with both path values accumulated in fp32: the error floor is absolute
        # (~1e-6), so on units of a few elements, whose largest value can be small by chance, it is measured against >= 1
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
        assert err < 1e-5, ("brownian", unit, st, ms, err)

for i in range(n_cases):
    for fn in (case_offset, case_pyramid, case_colored, case_brownian):
        state = rng.getstate()
        try:
            fn()
        except Exception as e:
            bad += 1
            print("FAIL", fn.__name__, i, type(e).__name__, str(e)[:400])
print("done, failures:", bad)
