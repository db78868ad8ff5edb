"""Noise generators on the MI355X vs the oracle's stages fed with the *specified* Philox draws.

The oracle's generator stages are pinned to the reference on injected draws (tests/test_oracle_golden.py::
test_noise_fixtures); here the same stages consume the normals/uniforms that the RNG specification
(oracle noise.py::philox_normal, host uniforms) assigns to each draw, and the HIP kernels must reproduce the
result.  Draw n of a generator owns Philox streams n*256 + k: k=0 base normal, k=1.. auxiliary normals,
k=255 uniforms."""

import math

import numpy as np
import pytest
import torch

from skr_oracle import noise as ON
from conftest import note_margin
from skrample_amd import _hip
from skrample_amd.common import Step
from skrample_amd.pytorch import noise as PN
from skrample_amd.pytorch._philox_host import philox_u32, uniform01

pytestmark = pytest.mark.gpu
TOL = 1e-5  # relative (inf-norm) tolerance for fp32 generator outputs
# Round 3 asserted 2e-5 (Pyramid, colorize) and 5e-5 (Colored) here without a measured number beside them.  Round 4 measured every
# comparison of the suite (profiles/r04_parity_margins.txt; float64 evaluations of the same inputs recorded next to the fp32 oracle):
#   Colored   device vs fp32 oracle: median 4.2e-7, MAX 8.9e-7 over 414 comparisons (device vs float64 5.8e-7, oracle vs float64 3.4e-7)
#   Pyramid   device vs fp32 oracle: MAX 3.8e-7 (LDS kernels), 3.5e-7 (any-shape kernels; vs float64 2.4e-6 where the oracle itself is 2.3e-6 off)
#   colorize_noise vs the REFERENCE's recorded outputs: MAX 3.0e-7;  LDS kernels vs the hipFFT route: 4.8e-7
# so every generator bar is the stated 1e-5 now (10x or more above the measured maxima); `no_further_from_exact` stays as the rule a
# wider bar would have to be justified by, and is asserted for the two longest pipelines anyway.
PYRAMID_TOL = TOL
COLORED_TOL = TOL
COLORIZE_TOL = TOL


@pytest.fixture(scope="module")
def dev():
    _hip.load()
    return torch.device("cuda:0")


def spec_normal(seed: int, stream: int, shape) -> torch.Tensor:
    return torch.from_numpy(ON.philox_normal(seed, stream, int(np.prod(shape)))).reshape(tuple(shape))


def rel(a: torch.Tensor, b: torch.Tensor, family: str | None = None, bar: float | None = None, exact: torch.Tensor | None = None) -> float:
    """relative inf-norm error.  With `family` the MEASURED value is recorded (conftest.note_margin); with `exact` -- the float64
    evaluation of the same inputs -- also how far the device result and the fp32 reference each sit from it."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = ((a - b).abs().max() / b.abs().max()).item()
    if family is not None:
        note_margin(family, "rel inf-norm error vs the fp32 oracle", err, bar)
        if exact is not None:
            e = exact.detach().cpu().double()
            note_margin(family, "device vs float64 evaluation", ((a - e).abs().max() / e.abs().max()).item(), bar)
            note_margin(family, "fp32 oracle vs float64 evaluation", ((b - e).abs().max() / e.abs().max()).item(), None)
    return err


def no_further_from_exact(got: torch.Tensor, ref32: torch.Tensor, exact: torch.Tensor, slack: float = 1.5, floor: float = 2e-6) -> bool:
    """the bar where an fp32 pipeline of many roundings (FFT, interpolation + normalisation) cannot meet 1e-5 against ANOTHER fp32
    pipeline: the device result is no further from the float64 evaluation of the same inputs than the reference's own fp32 result
    (x `slack`; `floor` covers references that happen to land almost exactly)"""
    g, r, e = (t.detach().cpu().double() for t in (got, ref32, exact))
    scale = e.abs().max()
    return ((g - e).abs().max() / scale).item() <= max(slack * ((r - e).abs().max() / scale).item(), floor)


def test_host_philox_matches_oracle_and_device(dev):
    seeds = np.array([1, 2**40 + 7, 2**64 - 3], dtype=np.uint64)
    host = philox_u32(seeds, 5 * 256 + 255, 3)
    for i, s in enumerate(seeds.tolist()):
        blocks = np.arange(3, dtype=np.uint64)
        ctr = np.stack([(blocks & 0xFFFFFFFF).astype(np.uint32), np.zeros(3, np.uint32), np.full(3, 5 * 256 + 255, np.uint32), np.zeros(3, np.uint32)], -1)
        key = np.array([s & 0xFFFFFFFF, s >> 32], dtype=np.uint32)
        assert np.array_equal(host[i], ON.philox4x32(ctr, key).reshape(-1))
        d = torch.empty(12, dtype=torch.int32, device=dev)
        _hip.check(_hip.load().skr_philox_u32(d.data_ptr(), s, 5 * 256 + 255, 0, 3, _hip.current_stream_ptr(dev)), "philox")
        assert np.array_equal(d.cpu().numpy().view(np.uint32), host[i])
    u = uniform01(seeds, 9, 6)
    assert ((u >= 0) & (u < 1)).all() and np.array_equal(u, u.astype(np.float32).astype(np.float64))


def test_random_generator(dev):
    seeds = [3, 4, 5]
    unit = (4, 16, 16)
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Random, unit, seeds, dtype=torch.float32)
    for n in range(3):
        got = g.generate(None)
        ref = torch.stack([spec_normal(s, n * 256, unit) for s in seeds])
        assert got.shape == (3, *unit) and (got.cpu() - ref).abs().max() < 4e-6
    one = PN.Random.from_inputs((5, 7), torch.Generator().manual_seed(77), dtype=torch.float32)
    assert (one.generate(None).cpu() - spec_normal(77, 0, (5, 7))).abs().max() < 4e-6
    big = PN.BatchTensorNoise.from_batch_inputs(PN.Random, (4, 128, 128), list(range(8)), dtype=torch.float32).generate(None)
    assert abs(big.mean().item()) < 5e-3 and abs(big.std().item() - 1) < 5e-3 and abs((big**4).mean().item() - 3) < 0.05


@pytest.mark.parametrize(("unit", "props"), [((4, 16, 16), PN.OffsetProps()), ((4, 33, 20), PN.OffsetProps(dims=(0, 2), strength=0.5)), ((16, 8, 8), PN.OffsetProps(dims=(1,), strength=1.5)), ((3, 5), PN.OffsetProps(dims=(-1,))), ((2, 3, 8, 8), PN.OffsetProps(dims=(0, 1))), ((4, 8, 16), PN.OffsetProps(dims=(-1,))), ((4, 8, 16), PN.OffsetProps(dims=(0, 2), strength=0.7)), ((2, 3, 2, 4, 8), PN.OffsetProps(dims=(0, 1), strength=0.4)), ((2, 3, 4, 2, 2, 8), PN.OffsetProps(dims=(1, 2, 5)))])
def test_offset(unit, props, dev):
    seeds = [11, 12]
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Offset, unit, seeds, props=props, dtype=torch.float32)
    nd = len(unit)
    dims = tuple(d + nd if d < 0 else d for d in props.dims)
    for n in range(2):
        got = g.generate(None).cpu()
        refs = []
        for s in seeds:
            draws = [spec_normal(s, n * 256 + 1, ON.offset_shape(unit, dims)), spec_normal(s, n * 256, unit)]
            refs.append(ON.offset_noise(unit, ON.Replay(draws).randn, dims, props.strength))
        assert rel(got, torch.stack(refs), "offset", TOL) < TOL, (unit, props, n)


@pytest.mark.parametrize("unit", [(4, 16, 16), (3, 5, 7)])
def test_brownian(unit, dev, monkeypatch):
    "increments of one fixed Brownian path per seed: oracle parity, additivity over adjacent steps, determinism"
    seeds = [31, 32]
    # default: ONE path per seed whatever the discretisation (the dyadic tree): a 20-step and a 40-step walk over [0.25, 0.5] add up to the
    # same W(0.5) - W(0.25), and a generator asked the coarse step first agrees with one asked the fine steps first
    mk = lambda: PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, props=PN.BrownianProps(), dtype=torch.float32)  # noqa: E731
    coarse, fine, whole = mk(), mk(), mk()
    w20 = sum(coarse.generate(Step.from_int(k, 20)).cpu().double() for k in range(5, 10)) * math.sqrt(1 / 20)
    w40 = sum(fine.generate(Step.from_int(k, 40)).cpu().double() for k in range(10, 20)) * math.sqrt(1 / 40)
    assert coarse._state["brownian_grid"] is None and fine._state["brownian_grid"] is None
    assert rel(w20, w40) < 1e-5 and rel(w20, whole.generate(Step(0.25, 0.5)).cpu().double() * 0.5) < 1e-5
    monkeypatch.setattr(PN, "BROWNIAN_SCHEDULE_PARTITION", True)  # the opt-in: the path over the schedule's own partition (one run, one step count)
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, props=PN.BrownianProps(), dtype=torch.float32)
    steps = [Step(0.35, 0.4), Step(0.4, 0.45), Step(0.35, 0.45), Step(0.0, 0.05), Step(0.95, 1.0), Step(0.5, 0.75)]
    got = {s: g.generate(s).cpu().double() for s in steps}
    assert g._state["brownian_grid"] == 20  # the first query was one cell of the 20-cell partition: the path is built over that partition
    for s, v in got.items():
        ref = torch.stack([ON.brownian_noise(seed, unit, s, grid=20) for seed in seeds])
        assert rel(v, ref, "brownian (own specification)", TOL) < TOL, s
    # a generator first asked an off-partition step keeps the dyadic tree; any later query is served from that one path
    d = PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, props=PN.BrownianProps(), dtype=torch.float32)
    for s in (Step(0.123, 0.777), Step(0.35, 0.4), Step(0.4, 0.45), Step(0.35, 0.45)):
        v = d.generate(s).cpu().double()
        assert d._state["brownian_grid"] is None
        assert rel(v, torch.stack([ON.brownian_noise(seed, unit, s) for seed in seeds]), "brownian (own specification)", TOL) < TOL, s
    # off-grid queries of a partition generator: inside a cell, and across cells
    for s in (Step(0.36, 0.39), Step(0.123, 0.777), Step(0.349, 0.4)):
        v = g.generate(s).cpu().double()
        assert rel(v, torch.stack([ON.brownian_noise(seed, unit, s, grid=20) for seed in seeds]), "brownian (own specification)", TOL) < TOL, s
    a, b, ab = got[steps[0]], got[steps[1]], got[steps[2]]
    assert rel((a + b) * math.sqrt(0.05), ab * math.sqrt(0.1)) < 1e-5  # W(.35,.4) + W(.4,.45) = W(.35,.45)
    assert torch.equal(g.generate(Step(0.4, 0.35)).cpu().double(), a)  # direction-normalised, stateless
    assert torch.equal(g.generate(Step(1.0, 1.05)).cpu().double(), got[steps[4]])  # clamped into [0,1]
    assert not torch.equal(a[0], a[1])
    white = g.generate(None)  # no step: plain white noise, a fresh draw each call
    assert white.shape == (2, *unit) and not torch.equal(white, g.generate(None))
    h = PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, props=PN.BrownianProps(), dtype=torch.bfloat16).generate(steps[0])
    assert h.dtype == torch.bfloat16 and (h.cpu().double() - a).abs().max() <= 2.0**-6
    # the W(time_to) cache: a query that starts where the previous one ended evaluates one path instead of two and
    # must produce the same bits as a fresh generator that evaluates both
    for dtype in (torch.float32, torch.bfloat16):
        seq = PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, dtype=dtype)
        chain = [seq.generate(Step.from_int(k, 20)) for k in range(5, 9)]  # 3 cache hits
        assert seq._state["brownian_cache_time"] == 9 / 20
        for k, v in zip(range(5, 9), chain):
            fresh = PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, dtype=dtype).generate(Step.from_int(k, 20))
            assert torch.equal(v, fresh), (dtype, k)
        assert torch.equal(seq.generate(Step.from_int(2, 20)), PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, dtype=dtype).generate(Step.from_int(2, 20)))  # miss after hits


def test_brownian_statistics_and_wrapper(dev):
    from skrample_amd import scheduling
    from skrample_amd.diffusers import SkrampleWrapperScheduler
    from skrample_amd.sampling import structured

    unit, seeds = (4, 64, 64), list(range(8))
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Brownian, unit, seeds, dtype=torch.float32)
    incs = [g.generate(Step.from_int(k, 20)).double() for k in range(20)]
    for v in incs[:4]:
        assert abs(v.mean().item()) < 8e-3 and abs(v.std().item() - 1) < 8e-3 and abs((v**4).mean().item() - 3) < 0.08
    for u, v in ((incs[0], incs[1]), (incs[3], incs[9]), (incs[18], incs[19])):
        assert abs((u * v).mean().item()) < 8e-3  # disjoint increments are independent
    total = sum(incs) * math.sqrt(1 / 20)  # = W(1), a unit normal
    assert abs(total.std().item() - 1) < 8e-3
    assert rel(total.float(), g.generate(Step(0.0, 1.0)).double().float()) < 1e-4

    w = SkrampleWrapperScheduler(structured.DPM(order=2, stochasticity=1.0), scheduling.Karras(scheduling.Scaled()), noise_type=PN.Brownian)
    w.set_timesteps(6)
    x = torch.randn(2, *unit, device=dev, dtype=torch.bfloat16)
    for t in w.timesteps:
        x = w.step(torch.randn_like(x), t, x, generator=[torch.Generator().manual_seed(3), torch.Generator().manual_seed(4)], return_dict=False)[0]
    assert torch.isfinite(x.float()).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32, torch.float64])
def test_vector_and_generic_kernels_agree_bitwise(dtype, dev):
    "the packed 8-wide kernels (aligned base) and the generic ones (base off by one element) must write the same bits"
    import ctypes

    lib, unit, batch = _hip.load(), (4, 8, 16), 3
    numel = 4 * 8 * 16
    seeds = torch.tensor([21, 22, 23], dtype=torch.int64, device=dev)
    hs = _hip.current_stream_ptr(seeds.device)
    code = _hip.DTYPE_CODE[dtype]
    for mask in (None, 1, 4, 5):
        buf = torch.zeros(2, batch * numel + 8, dtype=dtype, device=dev)
        for row, shift in ((0, 0), (1, 1)):
            ptr = buf[row, shift:].data_ptr()
            if mask is None:
                _hip.check(lib.skr_noise_random(ptr, code, seeds.data_ptr(), 512, batch, numel, hs), "skr_noise_random")
            else:
                shape = (ctypes.c_int64 * 3)(*unit)
                _hip.check(lib.skr_noise_offset(ptr, code, seeds.data_ptr(), 512, 513, batch, shape, 3, mask, 0.6, hs), "skr_noise_offset")
        torch.cuda.synchronize()
        fast, generic = buf[0, : batch * numel], buf[1, 1 : batch * numel + 1]
        assert torch.equal(fast, generic), (dtype, mask)
        assert fast.float().std().item() > 0.5 and (buf[0, batch * numel :] == 0).all() and (buf[1, 0] == 0).all()


def test_offset_static_and_dtypes(dev):
    unit, seeds = (4, 8, 8), [5]
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Offset, unit, seeds, props=PN.OffsetProps(static=True), dtype=torch.float32)
    a, b = g.generate(None).cpu(), g.generate(None).cpu()
    off = spec_normal(5, 1, (4, 1, 1)) * 0.2**2  # drawn once, at the first call
    assert rel(a, spec_normal(5, 0, unit) + off) < TOL and rel(b, spec_normal(5, 256, unit) + off) < TOL
    h = PN.BatchTensorNoise.from_batch_inputs(PN.Offset, unit, seeds, props=PN.OffsetProps(), dtype=torch.bfloat16).generate(None)
    ref = (spec_normal(5, 0, unit) + spec_normal(5, 1, (4, 1, 1)) * 0.04).bfloat16()
    assert h.dtype == torch.bfloat16 and (h.cpu().float() - ref.float()).abs().max() <= 2.0**-6


def test_pyramid_level_geometry():
    "vectorised host geometry == the oracle's level rule, sample by sample"
    seeds = np.arange(40, dtype=np.uint64) + 100
    u = uniform01(seeds, 3 * 256 + 255, PN.PYRAMID_MAX_LEVELS)
    for hw, resize_h in (((128, 128), True), ((256, 256), True), ((1, 64), False), ((33, 20), True)):
        table, counts = PN.pyramid_level_tables(hw, resize_h, u)
        for b in range(len(seeds)):
            it = iter(u[b].tolist())
            shape = (4, *hw) if resize_h else (4, hw[1])
            want = [run[-2:] if resize_h else (1, run[-1]) for _, run, _ in ON.pyramid_levels(shape, lambda: next(it), (-1, -2) if resize_h else (-1,))]
            assert counts[b] == len(want) and [tuple(r) for r in table[b, : counts[b]].tolist()] == [tuple(x) for x in want]


def pyramid_reference(unit, seed: int, stream: int, double: bool = False, **kw) -> torch.Tensor:
    "oracle pyramid fed the draws the specification assigns: base stream+0, level l stream+1+l, uniforms stream+255 (double: evaluated in float64)"
    uniforms = uniform01(np.array([seed], dtype=np.uint64), stream + 255, 8)[0].tolist()
    state = {"level": 0, "base_done": False}
    cast = (lambda t: t.double()) if double else (lambda t: t)

    def randn(shape):
        if not state["base_done"]:
            state["base_done"] = True
            return cast(spec_normal(seed, stream, shape))
        l = state["level"]
        state["level"] += 1
        return cast(spec_normal(seed, stream + 1 + l, shape))

    it = iter(uniforms)
    return ON.pyramid_noise(unit, randn, lambda: next(it), **kw)


@pytest.mark.parametrize(
    ("unit", "kw"),
    [((4, 16, 16), {}), ((4, 32, 24), {}), ((16, 16, 16), {}), ((4, 128, 128), {}), ((4, 256, 256), {}), ((2, 128, 512), {}), ((1, 400, 128), {}), ((4, 64, 64), dict(strength=0.6, depth=1)), ((8, 64), dict(dims=(-1,)))],  # (4-D unit shapes fail inside the reference itself)
)
def test_pyramid(unit, kw, dev):
    seeds = [21, 22, 23]
    props = PN.PyramidProps(**kw)
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=props, dtype=torch.float32)
    for n in range(2):
        got = g.generate(None).cpu()
        # the level geometry drawn on the device equals the host/oracle rule
        lead, h, w, resize_h = PN.Pyramid._geometry(unit, props)
        table, counts = PN.pyramid_level_tables((h, w), resize_h, uniform01(np.array(seeds, dtype=np.uint64), n * 256 + 255, PN.PYRAMID_MAX_LEVELS))
        dev_levels = g._state["levels"].cpu().numpy()
        dev_counts = dev_levels[len(seeds) * 16 :]
        assert np.array_equal(dev_counts, counts)
        for b in range(len(seeds)):
            assert np.array_equal(dev_levels[b * 16 : b * 16 + counts[b] * 2].reshape(-1, 2), table[b, : counts[b]])
        ref = torch.stack([pyramid_reference(unit, s, n * 256, **kw) for s in seeds])
        exact = torch.stack([pyramid_reference(unit, s, n * 256, double=True, **kw) for s in seeds])
        assert rel(got, ref, "pyramid (LDS kernels)", PYRAMID_TOL, exact) < PYRAMID_TOL, (unit, kw, n, rel(got, ref))
        assert (got.reshape(3, -1).std(dim=1) - 1).abs().max() < 1e-4


@pytest.mark.parametrize("unit", [(1, 256, 256), (1, 96, 640), (1, 30, 90)])
def test_pyramid_level_geometry_on_the_device_many_seeds(unit, dev):
    """the level sizes hinge on int(size / r**i): the device's r**i (squaring in double-double arithmetic) must be the double Python's
    float ** int gives, for every seed -- 6000 samples x 8 levels per shape, on the LDS kernels and on the any-shape geometry kernel"""
    seeds = list(range(7000, 13000))
    props = PN.PyramidProps()
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=props, dtype=torch.bfloat16)
    g.generate(None)
    lead, h, w, resize_h = PN.Pyramid._geometry(unit, props)
    table, counts = PN.pyramid_level_tables((h, w), resize_h, uniform01(np.array(seeds, dtype=np.uint64), 255, PN.PYRAMID_MAX_LEVELS))
    dev_levels = g._state["levels"].cpu().numpy()
    assert np.array_equal(dev_levels[len(seeds) * 16 :], counts)
    got = dev_levels[: len(seeds) * 16].reshape(len(seeds), 8, 2)
    keep = np.arange(8)[None, :, None] < counts[:, None, None]
    assert np.array_equal(np.where(keep, got, 0), np.where(keep, table, 0))


@pytest.mark.parametrize(("unit", "kw"), [((2, 30, 90), {}), ((3, 27, 18), dict(strength=0.6)), ((1, 400, 400), {}), ((2, 7), dict(dims=(-1,))), ((3, 50), dict(dims=(-1,), depth=1)), ((1, 4, 4), {})])
def test_pyramid_any_shape(unit, kw, dev):
    "widths that are not multiples of 4 and planes beyond the LDS level stage take the global-memory fallback; same oracle, same bar"
    seeds = [41, 42]
    props = PN.PyramidProps(**kw)
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=props, dtype=torch.float32)
    for n in range(2):
        got = g.generate(None).cpu()
        ref = torch.stack([pyramid_reference(unit, s, n * 256, **kw) for s in seeds])
        exact = torch.stack([pyramid_reference(unit, s, n * 256, double=True, **kw) for s in seeds])
        assert rel(got, ref, "pyramid (any-shape kernels)", PYRAMID_TOL, exact) < PYRAMID_TOL, (unit, kw, n, rel(got, ref))
    if unit != (1, 4, 4):
        assert "any_shape" in g._state
    h16 = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=props, dtype=torch.bfloat16).generate(None)
    ref = torch.stack([pyramid_reference(unit, s, 0, **kw) for s in seeds])
    assert (h16.cpu().float() - ref).abs().max() <= 2.0**-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize(
    ("unit", "dims"),
    [((4, 16, 24), (0, 1)), ((4, 16, 24), (0, 2)), ((6, 10, 12), (0, 2)), ((6, 10, 12), (0, 1)), ((4, 16, 24), (1, 2)), ((7, 33, 20), (-3, -1)), ((16, 64, 8), (0, 1))],
    # (units with two or more untouched axes fail inside the reference itself -- its un-permute assumes one; not pinned, not tested)
)
def test_pyramid_over_any_axis_pair(unit, dims, dev):
    """`dims` subsets other than the trailing axes (reference noise.py:146-193 permutes, interpolates slice by slice, permutes
    back; level normals are drawn in the unit's own axis order): the generator against the oracle -- itself pinned to the
    reference's outputs for (0,1) / (0,2) / (1,2) by tests/golden/noise_dims.npz"""
    seeds = [61, 62]
    props = PN.PyramidProps(dims=dims)
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=props, dtype=torch.float32)
    for n in range(2):
        got = g.generate(None).cpu()
        ref = torch.stack([pyramid_reference(unit, s, n * 256, dims=dims) for s in seeds])
        exact = torch.stack([pyramid_reference(unit, s, n * 256, double=True, dims=dims) for s in seeds])
        assert rel(got, ref, "pyramid (any axis pair)", PYRAMID_TOL, exact) < PYRAMID_TOL, (unit, dims, n, rel(got, ref))
        assert (got.reshape(len(seeds), -1).std(dim=1) - 1).abs().max() < 1e-4
    h16 = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=props, dtype=torch.bfloat16).generate(None)
    ref = torch.stack([pyramid_reference(unit, s, 0, dims=dims) for s in seeds])
    assert (h16.cpu().float() - ref).abs().max() <= 2.0**-6 * max(1.0, ref.abs().max().item())


def test_pyramid_dims_the_reference_rejects(dev):
    "a single non-trailing axis fails inside the reference's own permute (recorded in noise_dims.npz); three axes would be 5-D bicubic"
    for unit, dims in (((4, 16, 24), (0,)), ((4, 16, 24), (1,)), ((4, 8, 16, 24), (1, 2, 3)), ((4, 16, 24), (5,))):
        g = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, [1], props=PN.PyramidProps(dims=dims), dtype=torch.float32)
        with pytest.raises(PN.SkrampleHipError):
            g.generate(None)


def test_pyramid_static(dev):
    "PyramidProps.static: the pyramid component of the first draw is reused, only the base normal is fresh"
    unit, seeds = (4, 32, 32), [51, 52]
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Pyramid, unit, seeds, props=PN.PyramidProps(static=True), dtype=torch.float32)
    for n in range(3):
        got = g.generate(None).cpu()
        refs = []
        for s in seeds:
            uniforms = iter(uniform01(np.array([s], dtype=np.uint64), 255, 8)[0].tolist())
            level = {"l": 0}

            def lv(shape, s=s, level=level):
                v = spec_normal(s, 1 + level["l"], shape)
                level["l"] += 1
                return v

            pyr = ON.pyramid_component(unit, lv, lambda: next(uniforms))
            refs.append(ON.pyramid_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), None, static_pyramid=pyr))
        assert rel(got, torch.stack(refs), "pyramid (static)", PYRAMID_TOL) < PYRAMID_TOL, n


def test_pyramid_through_wrapper(dev):
    "cfg5-style use: RKUltra + Pyramid noise through the scheduler wrapper (noise realised as a tensor term)"
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS

    w = PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=2, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps())
    w.set_timesteps(3)
    x = torch.randn(2, 4, 32, 32, device=dev).bfloat16()
    for t in w.timesteps:
        x = w.step(torch.randn_like(x), t, x, generator=[1, 2], return_dict=False)[0]
    assert torch.isfinite(x.float()).all() and x.dtype == torch.bfloat16


@pytest.mark.parametrize("unit", [(4, 16, 16), (16, 16, 16), (1, 32, 64), (4, 128, 128), (16, 128, 128), (4, 64, 64), (16, 64, 64), (8, 32, 32), (2, 128, 64), (4, 32, 128), (128, 128), (1, 128, 128),
                                  (16, 96, 96), (4, 96, 128), (4, 128, 96), (4, 160, 96), (4, 80, 80), (96, 96), (4, 112, 144), (4, 104, 152), (2, 168, 96), (56, 88), (4, 28, 44), (8, 136, 120), (1, 124, 116), (2, 244, 68), (16, 13, 60, 104), (4, 5, 96, 96), (3, 96, 96), (12, 64, 64), (2, 3, 4, 40, 24), (5, 128, 128), (4, 90, 160), (2, 30, 40), (2, 6, 4), (4, 120, 4), (2, 2, 12), (2, 12), (102, 4), (2, 48, 96), (8, 24, 12), (4, 96, 160), (1, 192, 96), (4, 160, 160), (2, 192, 192), (160, 160), (192, 192), (2, 320, 64), (2, 256, 256), (16, 256, 128), (4, 128, 512), (2, 8, 256), (4, 96, 96), (16, 19, 13), (3, 40), (100,), (4, 1, 152, 104), (4, 256, 256), (256, 256), (2, 512, 64), (64, 64), (3, 4, 6, 8), (4, 5, 8, 16), (16, 3, 12, 10), (128, 3, 8, 8), (96, 2, 6, 10), (3, 4, 5, 6, 8), (2, 3, 2, 5, 4, 6), (5, 7, 3, 9, 10)])
def test_colored(unit, dev):
    seeds = [31, 32]
    cases = [
        (PN.ColoredProps(), [None, Step(0.0, 0.05), Step(0.45, 0.5), Step(0.95, 1.0)]),
        (PN.ColoredProps(energy=2.5, color_start=1.5, color_end=-3, color_curve=0), [Step(0.3, 0.4)]),
        (PN.ColoredProps(energy=-1.5, color_start=0.0), [None]),
    ]
    for props, steps in cases:
        g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=props, dtype=torch.float32)
        kw = dict(energy=props.energy, color_start=props.color_start, color_end=props.color_end, color_curve=props.color_curve)
        for n, st in enumerate(steps):
            got = g.generate(st).cpu()
            ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), st, **kw) for s in seeds])
            exact = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape).double(), st, **kw) for s in seeds])
            err = rel(got, ref, "colored", COLORED_TOL, exact)
            assert err < TOL and no_further_from_exact(got, ref, exact, slack=3.0), (unit, props, st, err)
            if props.energy is not None:
                assert (got.reshape(2, -1).std(dim=1) - abs(props.energy)).abs().max() < 1e-4


def test_colored_spectrum_slope(dev):
    "reference tests/self_noise.py:63-80: the measured PSD slope follows the requested exponent"
    from scipy import fft
    from scipy.stats import linregress

    def slope(data):
        F = fft.fftshift(fft.fftn(data))
        psd = np.abs(F) ** 2
        mesh = np.meshgrid(*[fft.fftshift(fft.fftfreq(s)) for s in data.shape], indexing="ij")
        r = np.sqrt(sum(m**2 for m in mesh))
        mask = r > 0
        rf, pf = r[mask], psd[mask]
        nb = min(data.shape) // 2
        edges = np.linspace(rf.min(), rf.max(), nb + 1)
        idx = np.digitize(rf, edges) - 1
        centers = 0.5 * (edges[:-1] + edges[1:])
        power = np.array([pf[idx == i].mean() if (idx == i).any() else 0 for i in range(nb)])
        ok = (power > 0) & (centers > 0)
        return -linregress(np.log(centers[ok]), np.log(power[ok])).slope

    for exponent in (-3, -1.5, 0, 1.5, 3):
        for unit in ((65536,), (1024, 1024), (128, 128, 128)):
            g = PN.Colored(unit, 5, torch.float32, PN.ColoredProps(color_curve=0, color_start=exponent, color_end=-exponent))
            assert abs(exponent - slope(g.generate(None).cpu().numpy())) < 0.1
            assert abs(-exponent - slope(g.generate(Step(0, 1)).cpu().numpy())) < 0.1


@pytest.mark.parametrize("unit", [(65536,), (1024, 1024), (128, 128, 128)])
def test_colored_energy(unit, dev):
    "reference tests/self_noise.py:83-103: the output std is |energy| exactly, or ~1 (the white noise's own) without it"
    rng = np.random.default_rng(3)
    for energy in (None, -3, -1.5, 0, 1.5, 3):
        props = PN.ColoredProps(energy=energy, color_start=float(rng.standard_normal()), color_end=float(rng.standard_normal()))
        g = PN.Colored(unit, 9, torch.float32, props)
        for st in (None, Step(0, 1)):
            std = g.generate(st).double().std().item()
            if energy is None:
                assert abs(1 - std) < 1e-2, (unit, energy, std)
            else:
                assert abs(abs(energy) - std) < 1e-5 * max(1.0, abs(energy)), (unit, energy, std)


def test_colored_with_unipc_wrapper(dev):
    "cfg3: UniPC-3 SDE + Colored noise through the scheduler wrapper vs the oracle fed the same realised noise"
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skr_oracle import samplers as OA
    from skr_oracle import schedules as OS
    from skr_oracle import wrapper as OW
    from skrample_amd.sampling import models as PM
    from skrample_amd.sampling import structured as PT

    steps, shape, seeds = 6, (2, 16, 16, 16), [41, 42]
    w = PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps())
    o = OW.StepDriver(OA.make("unipc", 3, eta=1), OS.linear(), "flow")
    w.set_timesteps(steps)
    o.set_timesteps(steps)
    shadow = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, shape[1:], seeds, props=PN.ColoredProps(), dtype=torch.float32)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(shape, generator=g)
    for i, t in enumerate(w.timesteps):
        out = torch.randn(shape, generator=g)
        got = w.step(out.to(dev), t, x.to(dev), generator=seeds, return_dict=False)[0]
        noise = shadow.generate(Step.from_int(i, steps)).cpu()  # the same draw the wrapper's generator made
        ref = o.step(out, t, x, noise=noise)[0]
        assert rel(got, ref, "colored weights / misc", TOL) < 1e-5, i
        x = ref


@pytest.mark.parametrize(
    ("kind", "props", "unit", "batch"),
    [
        (PN.Random, None, (4, 128, 128), 64),
        (PN.Offset, PN.OffsetProps(), (4, 128, 128), 64),
        (PN.Pyramid, PN.PyramidProps(), (4, 256, 256), 64),  # cfg5 per-GPU shard
        (PN.Colored, PN.ColoredProps(), (16, 128, 128), 32),  # cfg3 unit shape
    ],
)
def test_generators_full_size_properties(kind, props, unit, batch, dev):
    """BASELINE-size units: batch shards reproduce the full batch bit for bit (what makes 8-GPU sharding exact),
    repeated construction is deterministic, successive draws differ, moments are sane."""
    seeds = [1000 + i for i in range(batch)]
    step = Step(0.45, 0.5)

    def draws(sd, n=2):
        g = PN.BatchTensorNoise.from_batch_inputs(kind, unit, sd, props=props, dtype=torch.bfloat16)
        return [g.generate(step) for _ in range(n)]

    full = draws(seeds)
    again = draws(seeds)
    lo, hi = draws(seeds[: batch // 2]), draws(seeds[batch // 2 :])
    for f, a, l, h in zip(full, again, lo, hi):
        assert torch.equal(f, a)
        assert torch.equal(f[: batch // 2], l) and torch.equal(f[batch // 2 :], h)
    assert not torch.equal(full[0], full[1])
    x = full[0].float().reshape(batch, -1)
    assert torch.isfinite(x).all()
    assert (x.std(dim=1) - 1).abs().max() < (0.02 if kind is not PN.Offset else 0.05)
    assert x.mean(dim=1).abs().max() < (0.05 if kind is not PN.Offset else 0.15)  # Offset: mean of 4 channel offsets ~ N(0, 0.02)


@pytest.mark.parametrize(("steps", "begin", "schedule_name"), [(10, 5, "sinner"), (11, 6, "scaled")])
def test_diffusers_brownian(steps, begin, schedule_name, dev):
    "mirror of the reference's tests/self_sampling.py:503-537 (Euler-Maruyama, fp64 compute, begin index, one generator)"
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.sampling import models as PM
    from skrample_amd.sampling import structured as PT

    schedule = PS.Sinner(PS.Linear()) if schedule_name == "sinner" else PS.Scaled()
    wrapper = PD.SkrampleWrapperScheduler(sampler=PT.Euler(stochasticity=1), schedule=schedule, model=PM.DataModel(), compute_scale=torch.float64, noise_type=PN.Brownian)
    generator = torch.Generator().manual_seed(42)
    wrapper.set_timesteps(steps)
    begin *= wrapper.order
    wrapper.set_begin_index(begin)
    outs = []
    for t in wrapper.timesteps[begin:]:
        outs.append(wrapper.step(torch.randn([1, 16, 128], dtype=torch.float64, device=dev), t, torch.randn([1, 16, 128], dtype=torch.float64, device=dev), return_dict=False, generator=generator)[0])
    assert all(torch.isfinite(o).all() and o.dtype == torch.float64 for o in outs)
    assert wrapper._noise_generator is not None and len(wrapper._noise_generator.generators) == 1
    assert isinstance(wrapper._noise_generator.generators[0], PN.Brownian)


@pytest.mark.parametrize(("steps", "begin", "order", "schedule_name"), [(10, 5, 1, "sinner"), (11, 6, 2, "scaled"), (10, 6, 5, "scaled"), (11, 5, 6, "sinner"), (10, 5, 12, "scaled")])
def test_rku_brownian(steps, begin, order, schedule_name, dev):
    "mirror of the reference's tests/self_sampling.py:582-623 (RKUltra wrapper with Brownian noise)"
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.sampling import models as PM

    schedule = PS.Sinner(PS.Linear()) if schedule_name == "sinner" else PS.Scaled()
    wrapper = PD.RKUltraWrapperScheduler(schedule=schedule, sampler_order=order, stochasticity=1, model=PM.DataModel(), compute_scale=torch.float64, noise_type=PN.Brownian)
    generator = torch.Generator().manual_seed(42)
    wrapper.set_timesteps(steps)
    begin *= wrapper.order
    wrapper.set_begin_index(begin)
    for t in wrapper.timesteps[begin:]:
        out = wrapper.step(torch.randn([1, 16, 128], dtype=torch.float64, device=dev), t, torch.randn([1, 16, 128], dtype=torch.float64, device=dev), return_dict=False, generator=generator)[0]
        assert torch.isfinite(out).all()
    assert wrapper._noise_generator is not None and len(wrapper._noise_generator.generators) == 1
    assert isinstance(wrapper._noise_generator.generators[0], PN.Brownian)


def test_colorize_noise_equals_reference_fixtures(dev):
    """Colored.colorize_noise on the white draws the REFERENCE consumed must give the reference's own recorded output
    (tests/golden/noise.npz, written by tools/make_golden.py from the imported reference) -- reference in, reference out"""
    from conftest import load_npz

    fx = load_npz("noise.npz")
    tags = sorted(k.rsplit("/", 1)[0] for k in fx if k.endswith("/out") and k.startswith("colored"))
    assert len(tags) >= 10
    steps = [None, Step(0.0, 0.05), Step(0.45, 0.5), Step(0.95, 1.0)]
    for tag in tags:
        kind = tag.split("/")[0]
        if kind == "colored_energy":
            props, step = PN.ColoredProps(energy=2.5, color_start=1.5, color_end=-3, color_curve=0), Step(0.3, 0.4)
        else:
            props, step = PN.ColoredProps(), steps[int(kind[-1])]
        white = torch.from_numpy(fx[f"{tag}/normal0"]).to(dev)
        got = PN.Colored.colorize_noise(white, exponent=PN.colored_exponent(step, props), energy=props.energy)
        ref = torch.from_numpy(fx[f"{tag}/out"])
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert rel(got, ref, "colorize_noise vs reference-recorded outputs", COLORIZE_TOL) < COLORIZE_TOL, (tag, rel(got, ref))
    # API behaviour of the utility (reference noise.py:337-403)
    w = torch.randn(1, 8, 1, 12, device=dev)
    assert PN.Colored.colorize_noise(w) is w
    e = PN.Colored.colorize_noise(w, exponent=0.0, energy=2.0)
    assert abs(e.std().item() - 2.0) < 1e-5
    c = PN.Colored.colorize_noise(w.bfloat16(), exponent=1.0)
    assert c.shape == w.shape and c.dtype == torch.bfloat16
    assert abs(c.float().std().item() - w.bfloat16().float().std().item()) < 2e-2
    with pytest.raises(_hip.SkrampleHipError):
        PN.Colored.colorize_noise(torch.randn(2, 2, 2, 2, 2, 2, 2, device=dev), exponent=1.0)  # seven transform axes


def test_colorize_noise_with_up_to_six_axes_equals_reference_fixtures(dev):
    """tests/golden/colorize_nd.npz: the reference's colorize_noise on tensors with 4, 5 and 6 transform axes (the whole tensor is
    one sample: a batched video latent has five).  hipFFT takes the last three axes, every outer axis is a direct DFT kernel."""
    from conftest import load_npz

    fx = load_npz("colorize_nd.npz")
    tags = sorted({k.split("/")[0] for k in fx})
    assert len(tags) == 4
    for tag in tags:
        white = torch.from_numpy(fx[f"{tag}/white"]).to(dev)
        exponent, energy = fx[f"{tag}/args"].tolist()
        got = PN.Colored.colorize_noise(white, exponent=exponent, energy=None if math.isnan(energy) else energy).cpu()
        ref = torch.from_numpy(fx[f"{tag}/out"])
        assert got.shape == ref.shape and rel(got, ref, "colorize_noise vs reference-recorded outputs (4-6 axes)", COLORIZE_TOL) < COLORIZE_TOL, (tag, rel(got, ref))


def test_component_methods(dev):
    "Offset.offset() and Pyramid.pyramid(): the components alone, as the reference exposes them (noise.py:104-106, 146-200)"
    unit = (4, 16, 24)
    o = PN.Offset.from_inputs(unit, 77, PN.OffsetProps(dims=(0, 2), strength=0.5), dtype=torch.float32)
    for n in range(2):
        got = o.offset().cpu()
        assert got.shape == (4, 1, 24) and rel(got, spec_normal(77, n * 256 + 1, (4, 1, 24)) * 0.25) < TOL
    for kw in ({}, dict(strength=0.6, depth=1)):
        p = PN.Pyramid.from_inputs(unit, 78, PN.PyramidProps(**kw), dtype=torch.float32)
        for n in range(2):
            got = p.pyramid().cpu()
            uniforms = iter(uniform01(np.array([78], dtype=np.uint64), n * 256 + 255, 8)[0].tolist())
            level = iter(range(8))
            ref = ON.pyramid_component(unit, lambda shape: spec_normal(78, n * 256 + 1 + next(level), shape), lambda: next(uniforms), **kw)
            assert got.shape == unit and rel(got, ref, "pyramid component", PYRAMID_TOL) < PYRAMID_TOL, (kw, n, rel(got, ref))


@pytest.mark.parametrize("kind", ["colored_unipc", "pyramid_dpm", "offset_euler", "pyramid_rk", "colored_white_start"])
def test_noise_drawn_ahead_on_the_side_stream_is_value_identical(kind, dev):
    """The wrappers draw the next step's Pyramid / Offset noise on a side stream while this step's kernel runs
    (prefetch_noise=True; Colored is opted in here through _AHEAD_KINDS to cover a generator with symbolic white steps).  Same values, same draw numbering as drawing at the moment of use -- also when the caller
    does not ask for the guessed step next (out-of-order timestep, a fresh set_timesteps in mid-run)."""
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.sampling import models as PM
    from skrample_amd.sampling import structured as PT

    def make(prefetch):
        if kind == "colored_unipc":
            return PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps(), prefetch_noise=prefetch)
        if kind == "colored_white_start":  # exponent 0 throughout: plain white noise, which stays symbolic (drawn in the step kernel)
            return PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Colored, noise_props=PN.ColoredProps(color_start=0.0, color_end=0.0), prefetch_noise=prefetch)
        if kind == "pyramid_dpm":
            return PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), noise_type=PN.Pyramid, noise_props=PN.PyramidProps(), prefetch_noise=prefetch)
        if kind == "offset_euler":
            return PD.SkrampleWrapperScheduler(PT.Euler(stochasticity=0.7), PS.Scaled(), noise_type=PN.Offset, noise_props=PN.OffsetProps(), prefetch_noise=prefetch)
        return PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps(), prefetch_noise=prefetch)

    g = torch.Generator().manual_seed(5)
    shape, steps = (3, 4, 32, 32), 6
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)

    def run(prefetch, order):
        w = make(prefetch)
        w._AHEAD_KINDS = (PN.Pyramid, PN.Offset, PN.Colored)
        w.set_timesteps(steps)
        ts = list(w.timesteps)
        outs = [torch.randn(shape, generator=torch.Generator().manual_seed(100 + i)).bfloat16().to(dev) for i in range(len(ts))]
        x, results = x0, []
        for i in order(len(ts)):
            x = w.step(outs[i], ts[i], x, generator=[7, 8, 9], return_dict=False)[0]
            results.append(x.clone())
        if order is in_order:  # a second run on the same wrapper: nothing of the first run's ahead-draw may leak into it
            w.set_timesteps(steps)
            x = x0
            for i in range(2):
                x = w.step(outs[i], ts[i], x, generator=[7, 8, 9], return_dict=False)[0]
                results.append(x.clone())
        torch.cuda.synchronize()
        # draws consumed so far: a noise tensor drawn ahead and not yet asked for does not count (it is rewound if it is not wanted)
        used = None if w._noise_generator is None else (w._noise_ahead[3] if w._noise_ahead is not None else w._noise_generator._draws)
        return results, used, w

    in_order = lambda n: range(n)  # noqa: E731
    a, draws_a, wa = run(True, in_order)
    b, draws_b, _ = run(False, in_order)
    assert len(a) == len(b) and all(torch.equal(u, v) for u, v in zip(a, b)), kind
    assert draws_a == draws_b
    if kind != "pyramid_rk":  # (the RK wrapper walks its stages strictly in order)
        assert wa._noise_side is not None  # the side stream was really used
    if kind not in ("pyramid_rk", "colored_unipc"):  # (UniPC's corrector system is singular on a non-monotone step history)
        jumps = lambda n: [0, 1, 3, 2, 4]  # noqa: E731  step 3 is asked for when 2 was drawn ahead: dropped, counter rewound
        a, draws_a, _ = run(True, jumps)
        b, draws_b, _ = run(False, jumps)
        assert all(torch.equal(u, v) for u, v in zip(a, b)) and draws_a == draws_b, kind


def test_capture_after_steps_with_noise_drawn_ahead(dev):
    "a wrapper that has drawn noise ahead on its side stream can still be captured into a HIP graph afterwards (nothing is drawn ahead inside a capture)"
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.graphs import capture_sampling_loop
    from skrample_amd.sampling import structured as PT

    shape, steps, seeds = (3, 4, 32, 32), 5, [3, 4, 5]
    g = torch.Generator().manual_seed(16)
    net = lambda x, t: x * (0.5 + t / 2000)  # noqa: E731
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)
    mk = lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), noise_type=PN.Pyramid, prefetch_noise=True)  # noqa: E731

    def eager(w, x, n):
        w.set_timesteps(steps)
        for t in w.timesteps.tolist()[:n]:
            x = w.step(net(x, t), t, x, generator=seeds, return_dict=False)[0]
        return x

    want = eager(mk(), x0, steps)
    w = mk()
    eager(w, x0, 3)  # stops mid-run: the noise of step 3 is in flight on the side stream
    assert w._noise_ahead is not None and w._noise_done is not None
    loop = capture_sampling_loop(w, net, x0, steps, seeds=seeds)
    assert torch.equal(loop(x0), want)
    assert torch.equal(eager(w, x0, steps), want)  # and eager again after the capture


@pytest.mark.parametrize("unit", [(4, 96, 96), (16, 96, 96), (96, 96), (4, 96, 128), (4, 160, 96), (2, 80, 80), (8, 24, 12), (4, 160, 160), (2, 192, 192), (160, 160),
                                  (4, 112, 144), (4, 104, 152), (2, 168, 96), (56, 88), (4, 28, 44), (1, 124, 116), (8, 136, 120), (2, 184, 100), (2, 108, 104), (16, 36, 52), (2, 244, 68), (4, 132, 140), (252, 68), (4, 90, 160), (2, 30, 40), (1, 126, 100), (8, 18, 12), (2, 6, 4), (4, 120, 4), (2, 2, 12), (2, 12), (102, 4)])
def test_mixed_radix_planes_run_on_the_lds_kernels_and_agree_with_hipfft(unit, dev):
    """(Round 4: the same shapes through the library's own any-length transforms as a third route.)  Planes whose sides are a power of two (>= 4) times an odd factor up to 63 (latents of 768 / 896 / 1152 / 1216 / 1280 / 1344-pixel images ...) take the hand-written mixed-radix plane kernel
    (round 3) -- `skr_noise_colored` itself answers OK for them, where it used to refuse everything but powers of two -- and the
    result agrees with the hipFFT route (`skr_noise_colored_any`) on the same seeds, which the oracle tests above pin."""
    import ctypes

    lib = _hip.load()
    seeds, batch = [51, 52, 53], 3
    dims = [d for d in unit if d != 1]
    n = int(np.prod(dims))
    half = n // dims[-1] * (dims[-1] // 2 + 1)
    sd = PN.seeds_tensor(seeds, dev)
    outs = []
    for route in ("lds", "own", "hipfft"):
        spec = torch.empty(batch * half, dtype=torch.complex64, device=dev)
        scratch = torch.empty(batch * n, dtype=torch.float32, device=dev)
        partials = torch.empty(4 * batch * 256, dtype=torch.float64, device=dev)
        out = torch.empty((batch, *unit), dtype=torch.float32, device=dev)
        st = _hip.current_stream_ptr(dev)
        if route == "lds":
            d1, d2, d3 = ([1] + dims)[-3:]
            status = lib.skr_noise_colored(out.data_ptr(), _hip.F32, spec.data_ptr(), scratch.data_ptr(), partials.data_ptr(), 256, sd.data_ptr(), 512, batch, d1, d2, d3, 1.0, 0, 0.0, st)
        else:
            # the N-D transform proper (3-D units would otherwise take the plane kernels + a direct outer-axis DFT): the library's own
            # any-length kernels (skr_fft_own.hip, the default) or hipFFT
            assert lib.skr_set_tuning(b"fft_rank", 3) == 0 and lib.skr_set_tuning(b"hipfft", 1 if route == "hipfft" else 0) == 0
            before = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_execs")
            try:
                status = lib.skr_noise_colored_any(out.data_ptr(), _hip.F32, spec.data_ptr(), scratch.data_ptr(), partials.data_ptr(), sd.data_ptr(), 512, batch, len(dims), (ctypes.c_int32 * len(dims))(*dims), 1.0, 0, 0.0, st)
            finally:
                assert lib.skr_set_tuning(b"fft_rank", 0) == 0 and lib.skr_set_tuning(b"hipfft", -1) == 0
            after = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_execs")
            assert (after[0] - before[0], after[1] - before[1]) == ((0, 1) if route == "hipfft" else (1, 0)), (route, before, after)
        assert status == 0, (route, status)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert rel(outs[0], outs[2], "colored: LDS kernels vs the hipFFT route", COLORIZE_TOL) < COLORIZE_TOL, rel(outs[0], outs[2])
    assert rel(outs[1], outs[2], "colored: own any-length transforms vs the hipFFT route", COLORIZE_TOL) < COLORIZE_TOL, rel(outs[1], outs[2])
    assert abs(outs[0].std().item() - 1.0) < 0.05
    # a side with a factor the plane kernel does not handle is still refused there (and served by skr_noise_colored_any through the Python layer)
    spec = torch.empty(260 * 131, dtype=torch.complex64, device=dev)
    scratch = torch.empty(260 * 260, dtype=torch.float32, device=dev)
    out = torch.empty(260 * 260, dtype=torch.float32, device=dev)
    for d2, d3 in ((96, 260), (45, 96), (260, 64), (134, 64)):  # odd part 65 / 67 > 63; odd height
        assert lib.skr_noise_colored(out.data_ptr(), _hip.F32, spec.data_ptr(), scratch.data_ptr(), partials.data_ptr(), 256, sd.data_ptr(), 512, 1, 1, d2, d3, 1.0, 0, 0.0, _hip.current_stream_ptr(dev)) == 7


@pytest.mark.parametrize("unit", [(4, 256, 256), (2, 256, 256), (4, 512, 512), (4, 256, 512), (2, 512, 256)])
def test_large_planes_with_few_channels_keep_their_middle_passes_in_one_tile(unit, dev, monkeypatch):
    """planes too large for the fused plane kernels under 2 or 4 channels: column transform, channel axis + radial weights and inverse column transform run
    in one tile residency (colored_mid_axes) -- the same per-line arithmetic as the three separate passes (SKR_COLORED_NO_MID=1), so the same bits; and the
    draw equals the oracle's rfftn route on the same white field as every other shape does"""
    outs = {}
    for mode in ("fused", "separate"):
        if mode == "separate":
            monkeypatch.setenv("SKR_COLORED_NO_MID", "1")
        for dtype in (torch.float32, torch.bfloat16):
            g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, [71, 72, 73], props=PN.ColoredProps(), dtype=dtype)
            outs[mode, dtype] = [g.generate(st).clone() for st in (None, Step(0.3, 0.4))]
    monkeypatch.delenv("SKR_COLORED_NO_MID", raising=False)
    for dtype in (torch.float32, torch.bfloat16):
        for a, b in zip(outs["fused", dtype], outs["separate", dtype]):
            assert torch.equal(a, b), (unit, dtype)
    assert abs(outs["fused", torch.float32][0].std().item() - 1.0) < 0.02


@pytest.mark.parametrize("unit", [(4, 96, 96), (16, 96, 96), (96, 96), (2, 160, 160), (160, 160), (2, 192, 192)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_compile_time_geometry_planes_agree_with_the_runtime_kernel(unit, dtype, dev):
    """96 / 160 / 192-point squares run an instantiation with its geometry folded in; SKR_FFT_NO_CONST_SIDE=1 takes the runtime-geometry kernel.
    Same algorithm, same operation order -- but the compiler contracts multiply-adds differently around constants, so the two agree to fp32
    rounding (a 16-bit result differs in the last place for well under 1 % of the elements), not bit for bit."""
    import os

    outs = []
    for const in (True, False):
        os.environ.pop("SKR_FFT_NO_CONST_SIDE", None)
        if not const:
            os.environ["SKR_FFT_NO_CONST_SIDE"] = "1"
        try:
            g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, [61, 62, 63], props=PN.ColoredProps(), dtype=dtype)
            outs.append([g.generate(st).clone() for st in (None, Step(0.3, 0.4))])
        finally:
            os.environ.pop("SKR_FFT_NO_CONST_SIDE", None)
    for a, b in zip(*outs):
        a32, b32 = a.float(), b.float()
        if dtype == torch.float32:
            assert rel(a32.cpu(), b32.cpu()) < 1e-5
        else:
            assert (a32 != b32).float().mean().item() < 0.01
            assert ((a32 - b32).abs() <= 2.0**-7 * torch.maximum(a32.abs(), b32.abs()) + 1e-6).all()
        assert torch.isfinite(a32).all() and abs(a32.std().item() - 1.0) < 0.05


@pytest.mark.parametrize("hipfft", [0, 1])
@pytest.mark.parametrize("unit", [(4, 96, 96), (16, 19, 13), (3, 40), (3, 4, 6, 8), (16, 2, 8, 8), (3, 4, 5, 6, 8), (2, 3, 2, 5, 4, 6)])
def test_colored_with_only_the_last_axis_on_hipfft(unit, hipfft, dev):
    """The route taken when a multi-dimensional hipFFT plan fails its self-check (rocFFT 7.2 can return a wrong real 2-D / 3-D
    plan in a process that has made many others -- tools/fft_probe.py): the last axis on a 1-D plan, every other axis on
    the direct-DFT kernels.  Forced here through skr_set_tuning("fft_rank", 1) and held to the same oracle bar -- with the last axis on
    the library's own transform (hipfft = 0, the default since round 4) and on hipFFT."""
    lib = _hip.load()
    assert lib.skr_set_tuning(b"fft_rank", 1) == 0 and lib.skr_set_tuning(b"hipfft", hipfft) == 0
    try:
        seeds = [31, 32]
        g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
        for n, st in enumerate((None, Step(0.45, 0.5))):
            got = g.generate(st).cpu()
            ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), st) for s in seeds])
            exact = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape).double(), st) for s in seeds])
            err = rel(got, ref, "colored (last axis on %s, others direct DFT)" % ("hipFFT" if hipfft else "the own transform"), COLORED_TOL, exact)
            assert err < TOL and no_further_from_exact(got, ref, exact, slack=3.0), (unit, st, err)
    finally:
        assert lib.skr_set_tuning(b"fft_rank", 0) == 0 and lib.skr_set_tuning(b"hipfft", -1) == 0


@pytest.mark.parametrize("hipfft", [0, 1])
@pytest.mark.parametrize("unit", [(5, 19, 13), (3, 30, 90), (6, 97, 14), (3, 4, 6, 10), (2, 3, 5, 6, 9)])
def test_colored_with_two_axes_on_the_transform_and_the_rest_direct(unit, hipfft, dev):
    """skr_set_tuning("fft_rank", 2): the last two axes on the N-D transform, every axis in front of them on the direct-DFT kernels,
    which then carry the radial weights -- so the own route must run BOTH of its axes (it skipped the outer one of the two
    whenever the unit had at most three axes, leaving that axis untransformed under the weights).  Awkward lengths, own route and hipFFT,
    same oracle bar as every other route."""
    lib = _hip.load()
    assert lib.skr_set_tuning(b"fft_rank", 2) == 0 and lib.skr_set_tuning(b"hipfft", hipfft) == 0
    try:
        seeds = [41, 42]
        before = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_execs")
        g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
        for n, st in enumerate((None, Step(0.45, 0.5))):
            got = g.generate(st).cpu()
            ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), st) for s in seeds])
            exact = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape).double(), st) for s in seeds])
            err = rel(got, ref, "colored (last two axes on %s, others direct DFT)" % ("hipFFT" if hipfft else "the own transforms"), COLORED_TOL, exact)
            assert err < TOL and no_further_from_exact(got, ref, exact, slack=3.0), (unit, st, err)
        after = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_execs")
        assert (after[0] - before[0] > 0, after[1] - before[1] > 0) == ((False, True) if hipfft else (True, False)), (before, after)
    finally:
        assert lib.skr_set_tuning(b"fft_rank", 0) == 0 and lib.skr_set_tuning(b"hipfft", -1) == 0


def _short_length(d: int) -> bool:
    "an axis length the tile transforms of skr_fft_own.hip take: anything up to 2048 (Bluestein), and up to 4096 the direct ones -- 2^a (a >= 1) times at most three factors out of 3, 5, 7, 11, 13"
    if d <= 2048:
        return True
    odd = 0
    for f in (13, 11, 7, 5, 3):
        while d % f == 0 and odd < 3:
            d, odd = d // f, odd + 1
    return d <= 4096 and d & (d - 1) == 0 and (odd == 0 or d >= 2)


def _own_length(d: int, last: bool = False) -> bool:
    "... or, as the LAST axis, an even length 2 A B with A, B such lengths (the half-length transform in four steps)"
    if _short_length(d):
        return True
    h = d // 2
    return last and d % 2 == 0 and any(h % a == 0 and _short_length(a) and _short_length(h // a) for a in range(2, int(h**0.5) + 1))


@pytest.mark.parametrize("unit", [(4, 97, 97), (4, 30, 90), (3, 250, 250), (2, 66, 130), (1, 45, 96), (2, 134, 64), (3, 7, 11, 13), (2, 1025), (1, 3, 2050), (4, 720, 1280), (2, 3, 5), (5, 1300),
                                  (2, 250, 60), (3, 3072), (3, 2560), (2, 90, 18), (1, 720, 30), (2, 3000), (2, 1500), (1, 6, 10), (3, 1080), (3, 3840),
                                  (2, 154, 182), (3, 2002), (2, 28, 44), (1, 4004), (2, 14, 22, 26), (16, 66, 130)])
def test_awkward_shapes_run_on_the_own_transforms(unit, dev):
    """odd sides, widths that are not multiples of 4, odd parts beyond 63, primes, lengths next to a power of two, a 720 x 1280 plane, one-,
    two- and three-level 2^a 3^b 5^c lengths (30, 60, 90, 250, 720, 3072 ...) and ones with four odd factors (1080, 1500: Bluestein again): every
    axis length up to 2048, and the direct ones up to 4096, run on skr_fft_own.hip -- no hipFFT plan is made, no hipFFT transform
    runs -- and meet the oracle bar; only a longer axis (2050, 3000) is still served by hipFFT"""
    lib = _hip.load()
    seeds = [61, 62]
    before = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_plans"), lib.skr_stat(b"hipfft_execs")
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
    for n, st in enumerate((None, Step(0.45, 0.5))):
        got = g.generate(st).cpu()
        ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), st) for s in seeds])
        exact = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape).double(), st) for s in seeds])
        err = rel(got, ref, "colored (own any-length transforms)", COLORED_TOL, exact)
        assert err < TOL and no_further_from_exact(got, ref, exact, slack=3.0), (unit, st, err)
    after = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_plans"), lib.skr_stat(b"hipfft_execs")
    assert all(_own_length(d, i == len(unit) - 1) for i, d in enumerate(unit))
    # (0 own transforms: a shape the LDS plane kernels take after all -- still no vendor transform)
    assert after[0] - before[0] in (0, 2) and after[1:] == before[1:], (unit, before, after)


@pytest.mark.parametrize("unit", [(65536,), (2, 8192), (3, 5000), (2, 3, 6000), (1, 10010), (2, 4100), (4, 12, 4098)])
def test_long_last_axes_run_on_the_own_transforms(unit, dev):
    """(Round 5.)  A last axis beyond one LDS tile -- 8192, 65536 samples of a waveform, 5000, 10010 ... -- is a half-length complex transform
    in four steps over the same tile kernels (n = 2 A B): no hipFFT plan, no hipFFT transform, the oracle bar as everywhere."""
    lib = _hip.load()
    seeds = [71, 72]
    before = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_plans"), lib.skr_stat(b"hipfft_execs")
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
    for n, st in enumerate((None, Step(0.45, 0.5))):
        got = g.generate(st).cpu()
        ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), st) for s in seeds])
        exact = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape).double(), st) for s in seeds])
        err = rel(got, ref, "colored (long last axis, four-step)", COLORED_TOL, exact)
        assert err < TOL and no_further_from_exact(got, ref, exact, slack=3.0), (unit, st, err)
    after = lib.skr_stat(b"own_fft_execs"), lib.skr_stat(b"hipfft_plans"), lib.skr_stat(b"hipfft_execs")
    assert after[0] - before[0] == 2 and after[1:] == before[1:], (unit, before, after)


def test_the_vendor_fft_runs_only_when_asked_for(dev):
    "a shape the own transforms do not take (a long axis that is not the last one; a last axis 2 x prime): refused -- and served once hipFFT is asked for"
    lib = _hip.load()
    for unit in ((4100, 6), (2, 2 * 4099)):
        g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, [5, 6], props=PN.ColoredProps(), dtype=torch.float32)
        plans = lib.skr_stat(b"hipfft_plans")
        with pytest.raises(_hip.SkrampleHipError):
            g.generate(Step(0.45, 0.5))
        assert lib.skr_stat(b"hipfft_plans") == plans
        assert lib.skr_set_tuning(b"hipfft", 1) == 0
        try:
            got = g.generate(Step(0.45, 0.5)).cpu()
        finally:
            assert lib.skr_set_tuning(b"hipfft", -1) == 0
        assert lib.skr_stat(b"hipfft_plans") > plans
        ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, 256, shape), Step(0.45, 0.5)) for s in (5, 6)])
        assert rel(got, ref, "colored (hipFFT on request)", COLORED_TOL) < TOL


@pytest.mark.parametrize("unit", [(2, 3, 2, 2, 3, 4, 6), (2, 2, 3, 2, 2, 2, 4, 4), (2, 1, 2, 2, 2, 2, 2, 2, 2, 2, 3, 4)])
def test_colored_with_seven_and_more_transform_axes(unit, dev, monkeypatch):
    """reference noise.py:373-403 transforms over every non-singleton axis, however many: the leading ones (up to nine) are direct-DFT kernels.
    (The reference's own torch.fft.rfftn stops at seven axes on the host -- MKL refuses more -- so for eight and twelve axes the oracle's
    two library calls are served axis by axis here: the same transform, composed.)"""
    if sum(d != 1 for d in unit) > 7:
        def rfftn(x):
            y = torch.fft.rfft(x, dim=-1)
            for d in range(x.dim() - 2, -1, -1):
                y = torch.fft.fft(y, dim=d)
            return y

        def irfftn(y, s):
            for d in range(y.dim() - 1):
                y = torch.fft.ifft(y, dim=d)
            return torch.fft.irfft(y, n=s[-1], dim=-1)

        monkeypatch.setattr(torch.fft, "rfftn", rfftn)
        monkeypatch.setattr(torch.fft, "irfftn", irfftn)
    seeds = [81, 82]
    g = PN.BatchTensorNoise.from_batch_inputs(PN.Colored, unit, seeds, props=PN.ColoredProps(), dtype=torch.float32)
    for n, st in enumerate((None, Step(0.45, 0.5))):
        got = g.generate(st).cpu()
        ref = torch.stack([ON.colored_noise(unit, lambda shape, s=s: spec_normal(s, n * 256, shape), st) for s in seeds])
        assert rel(got, ref, "colored (7+ axes)", COLORED_TOL) < TOL, (unit, st)



