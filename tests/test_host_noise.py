"""Structured noise on HOST-resident latents (skrample_amd/pytorch/host_noise.py): the package's own torch-CPU evaluation of
Offset / Pyramid / Colored, drawn from the caller's CPU generators in the reference's order.  The reference recorded
tests/golden/noise.npz and noise_dims.npz from generators seeded 11..20 and 40..45 (tools/make_golden.py::noise), together
with every draw it consumed: seeded alike, the host generators must consume the same draws and return the same bits.
No GPU, no oracle import in the product path (tests/test_boundary.py checks the latter for every module)."""

import numpy as np
import pytest
import torch

import skrample_amd.diffusers as PD
import skrample_amd.scheduling as PS
from skrample_amd.common import Step
from skrample_amd.pytorch import host_noise as HN
from skrample_amd.pytorch import noise as PN
from skrample_amd.sampling import structured as PT

from conftest import load_npz

UNITS = ((4, 16, 16), (4, 32, 24), (16, 16, 16))
STEPS = (None, Step(0.0, 0.05), Step(0.45, 0.5), Step(0.95, 1.0))


def one(kind, unit, seed, props=None, step=None):
    batch = HN.HostStructuredBatch(kind, unit, [torch.Generator().manual_seed(seed)], props)
    return batch.generate(step)[0]


@pytest.mark.parametrize("unit", UNITS)
def test_reference_recorded_generators_from_their_seeds(unit):
    fx = load_npz("noise.npz")
    u = "x".join(map(str, unit))
    cases = {
        f"offset/{u}": (PN.Offset, 11, None, None),
        f"offset_d02/{u}": (PN.Offset, 12, PN.OffsetProps(dims=(0, 2), strength=0.5), None),
        f"pyramid/{u}": (PN.Pyramid, 13, None, None),
        f"pyramid_depth1/{u}": (PN.Pyramid, 14, PN.PyramidProps(strength=0.6, depth=1), None),
        f"colored_energy/{u}": (PN.Colored, 20, PN.ColoredProps(energy=2.5, color_start=1.5, color_end=-3, color_curve=0), Step(0.3, 0.4)),
    }
    for j, st in enumerate(STEPS):
        cases[f"colored{j}/{u}"] = (PN.Colored, 15 + j, None, st)
    for tag, (kind, seed, props, step) in cases.items():
        got = one(kind, unit, seed, props, step)
        assert got.dtype == torch.float32 and torch.equal(got, torch.from_numpy(fx[f"{tag}/out"])), tag
    assert torch.equal(HN._radial_frequencies(unit, torch.device("cpu")), torch.from_numpy(fx[f"radial/{u}"]))


def test_the_draws_consumed_are_the_recorded_ones():
    "same order, same shapes, same values as the reference consumed (Pyramid: base normal, then per level a uniform and a reduced normal)"
    fx = load_npz("noise.npz")
    tag = "pyramid/4x32x24"
    seen_n, seen_u = [], []

    class Spy(HN._Draws):
        def normal(self, shape):
            v = super().normal(shape)
            seen_n.append(v.clone())
            return v

        def uniform(self):
            v = super().uniform()
            seen_u.append(v)
            return v

    batch = HN.HostStructuredBatch(PN.Pyramid, (4, 32, 24), [torch.Generator().manual_seed(13)], None)
    batch._sources = [Spy(batch.generators[0], torch.float32)]
    batch.generate(None)
    assert seen_u == fx[f"{tag}/uniforms"].tolist()
    assert len(seen_n) == int(fx[f"{tag}/n_normals"])
    for i, v in enumerate(seen_n):
        assert torch.equal(v, torch.from_numpy(fx[f"{tag}/normal{i}"]))


def test_pyramid_over_other_axis_pairs_and_the_choices_the_reference_rejects():
    fx = load_npz("noise_dims.npz")
    for unit in ((4, 16, 24), (6, 10, 12)):
        u = "x".join(map(str, unit))
        for k, dims in enumerate(((0, 1), (0, 2), (0,), (1,), (-2,), (1, 2))):
            tag = "pyramid_d" + "".join(str(d % len(unit)) for d in dims) + f"/{u}"
            if f"{tag}/out" in fx:
                got = one(PN.Pyramid, unit, 40 + k, PN.PyramidProps(dims=dims))
                assert torch.equal(got, torch.from_numpy(fx[f"{tag}/out"])), tag
            else:
                assert str(fx[f"{tag}/reference_error"]) == "RuntimeError"
                with pytest.raises(RuntimeError):
                    one(PN.Pyramid, unit, 40 + k, PN.PyramidProps(dims=dims))


def test_static_components_are_frozen_when_the_generator_is_built():
    g = torch.Generator().manual_seed(5)
    b = HN.HostStructuredBatch(PN.Offset, (3, 8, 8), [g], PN.OffsetProps(static=True))
    first, second = b.generate(None)[0], b.generate(None)[0]
    ref = torch.Generator().manual_seed(5)
    off = torch.randn((3, 1, 1), generator=ref) * 0.2**2  # drawn at construction, before any full-size normal
    assert torch.equal(first, torch.randn((3, 8, 8), generator=ref) + off)
    assert torch.equal(second, torch.randn((3, 8, 8), generator=ref) + off)
    p = HN.HostStructuredBatch(PN.Pyramid, (2, 8, 8), [torch.Generator().manual_seed(6)], PN.PyramidProps(static=True))
    a, c = p.generate(None)[0], p.generate(None)[0]
    assert not torch.equal(a, c) and abs(float(a.std()) - 1) < 1e-5 and abs(float(c.std()) - 1) < 1e-5


@pytest.mark.parametrize("kind,props", [(PN.Offset, None), (PN.Pyramid, PN.PyramidProps()), (PN.Colored, PN.ColoredProps())])
def test_wrapper_on_cpu_latents_with_structured_noise(kind, props):
    """`swap the import` on CPU tensors: a stochastic sampler with a structured generator steps host-resident latents; the noise the
    wrapper consumed is the stack of per-item draws of generators seeded like the caller's"""
    shape, steps = (2, 4, 16, 16), 5
    w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), noise_type=kind, noise_props=props)
    w.set_timesteps(steps)
    gens = [torch.Generator().manual_seed(100 + i) for i in range(shape[0])]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(shape, generator=g)
    shadow = HN.HostStructuredBatch(kind, shape[1:], [torch.Generator().manual_seed(100 + i) for i in range(shape[0])], props)
    w2 = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()))  # white noise: a different trajectory
    w2.set_timesteps(steps)
    x2 = x
    for i, t in enumerate(w.timesteps):
        out = torch.randn(shape, generator=g)
        nxt = w.step(out, t, x, generator=gens, return_dict=False)[0]
        assert nxt.shape == x.shape and nxt.dtype == x.dtype and torch.isfinite(nxt).all()
        x = nxt
    assert isinstance(w._noise_generator, HN.HostStructuredBatch) and w._noise_generator._draws == steps
    # the generator's stream is exactly the per-item streams, in order
    again = HN.HostStructuredBatch(kind, shape[1:], [torch.Generator().manual_seed(100 + i) for i in range(shape[0])], props)
    for i in range(steps):
        a, b = again.generate(Step.from_int(i, steps)), shadow.generate(Step.from_int(i, steps))
        assert torch.equal(a, b) and a.shape == shape


def test_brownian_on_host_tensors_says_what_is_missing():
    w = PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Scaled(), noise_type=PN.Brownian, noise_props=PN.BrownianProps())
    w.set_timesteps(3)
    x = torch.randn(1, 4, 8, 8)
    with pytest.raises(Exception, match="torchsde"):
        w.step(x, w.timesteps[0], x, generator=torch.Generator().manual_seed(1))


def test_a_unit_of_unit_axes_is_refused_with_torchs_own_error():
    "noise.py:371-377: squeeze() leaves no axis and torch.fft.rfftn refuses -- the same error here (live sweep, noise mode, seed 3000165)"
    with pytest.raises(RuntimeError, match="rfftn must transform at least one axis"):
        one(PN.Colored, (1, 1, 1), 3, PN.ColoredProps(energy=0.5, color_start=-1, color_end=-2, color_curve=2), Step(0.3, 0.9))
