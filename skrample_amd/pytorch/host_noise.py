"""Structured noise for HOST-resident latents (CPU torch tensors): the package's own torch-CPU evaluation of
Offset / Pyramid / Colored, so that a caller who keeps latents on the CPU -- the reference's generic path and BASELINE
config 1 -- can swap the import with any noise type, not only `Random`.

This is host plumbing, not the product path: device tensors never come here (`diffusers.py::_make_noise_generator`
routes by residency) and nothing under `oracle/` is imported.  Randomness is the CALLER's: one `torch.Generator` per
batch item, consumed in exactly the order the reference consumes it (skrample/pytorch/noise.py: Offset :98-113 draws
the offset, then the full normal; Pyramid :146-207 draws the full normal, then per level one uniform and one reduced
normal; Colored :407-425 draws one normal) -- seeded alike, the results are the reference's own bit for bit
(tests/test_host_noise.py replays tests/golden/noise.npz and noise_dims.npz from their generator seeds).

Brownian is the reference's torchsde tree; without that package there is nothing to stand in for it on the host.
"""

from __future__ import annotations

import math

import torch

from .._hip import SkrampleHipError
from ..common import Step
from . import noise as device_noise
from .noise import ColoredProps, OffsetProps, PyramidProps, colored_exponent


class _Draws:
    "one batch item's source of randomness: the caller's generator, drawn in fp32 on its own device"

    def __init__(self, generator: torch.Generator, dtype: torch.dtype):
        self.generator, self.dtype = generator, dtype

    def normal(self, shape) -> torch.Tensor:
        return torch.randn(tuple(shape), generator=self.generator, dtype=self.dtype, device=self.generator.device)

    def uniform(self) -> float:
        return torch.rand([1], generator=self.generator, dtype=self.dtype, device=self.generator.device).item()


# ---- Offset ---------------------------------------------------------------------------------------------------------
def offset_component(draws: _Draws, unit: tuple[int, ...], props: OffsetProps) -> torch.Tensor:
    "strength^2 * N over the axes listed in `dims` (size 1 elsewhere: broadcast by the sum)"
    kept = tuple(size if axis in props.dims else 1 for axis, size in enumerate(unit))
    return draws.normal(kept) * props.strength**2


# ---- Pyramid --------------------------------------------------------------------------------------------------------
def _resized_axes(unit: tuple[int, ...], dims) -> list[int]:
    nd = len(unit)
    axes = sorted({d + nd if d < 0 else d for d in dims})
    if not axes or any(not 0 <= a < nd for a in axes) or len(axes) > 3:
        raise ValueError(f"Pyramid dims {tuple(dims)} do not name one to three axes of a {nd}-axis unit")
    return axes


def _upsample(level: torch.Tensor, unit: tuple[int, ...], axes: list[int]) -> torch.Tensor:
    """`level` (the unit's axis order, reduced sizes on `axes`) brought to the unit's size by torch's linear / bilinear
    interpolation (align_corners=False) over the resized axes, every slice along the others on its own.  One batched
    interpolate call over all slices: per (slice, channel) it is the same arithmetic as a call per slice."""
    mode = ("linear", "bilinear", "bicubic")[len(axes) - 1]
    others = [a for a in range(len(unit)) if a not in axes]
    if len(others) > 1:
        # the reference folds the other axes into ONE leading axis and then permutes back with the unit's full rank
        # (noise.py:176-191): with two or more of them that permute raises -- same error class here, on the host path
        raise RuntimeError(f"Pyramid dims leave {len(others)} untouched axes in a {len(unit)}-axis unit: the reference's own permute fails there (noise.py:186-191)")
    moved = level.permute(*others, *axes)  # slices first, resized axes last
    lead = math.prod(moved.shape[: len(others)])
    target = tuple(unit[a] for a in axes)
    planes = moved.reshape(lead, 1, *moved.shape[len(others) :])
    grown = torch.nn.functional.interpolate(planes, target, mode=mode)
    grown = grown.reshape(*[unit[a] for a in others], *target)
    back = [0] * len(unit)
    for position, axis in enumerate([*others, *axes]):
        back[axis] = position
    return grown.permute(*back)


def pyramid_component(draws: _Draws, unit: tuple[int, ...], props: PyramidProps) -> torch.Tensor:
    "sum over levels >= skip of strength^l * upsample(N(level shape)); the sizes shrink by (2 + 2u)^l per level, cumulatively"
    axes = _resized_axes(unit, props.dims)
    running = list(unit)
    layers: list[torch.Tensor] = []
    for level in range(99):
        ratio = draws.uniform() * 2 + 2
        for a in axes:
            running[a] = max(1, int(running[a] / ratio**level))
        layers.append(_upsample(draws.normal(running), unit, axes).reshape(unit) * props.strength**level)
        if any(running[a] <= 1 for a in axes):
            break
    deepest = len(layers) - 1
    first = min(deepest, max(0, deepest - props.depth))
    total = torch.zeros(unit, dtype=draws.dtype, device=draws.generator.device)
    return total + sum(layers[first:])


# ---- Colored --------------------------------------------------------------------------------------------------------
def _radial_frequencies(shape: tuple[int, ...], device) -> torch.Tensor:
    "normalised distance from DC of every rfftn bin of a `shape` transform (half spectrum on the last axis)"
    axes = [torch.fft.fftfreq(n, d=1.0, device=device).abs() for n in shape[:-1]]
    axes.append(torch.arange(shape[-1] // 2 + 1, device=device) / shape[-1])
    radius = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).norm(p=2, dim=-1)
    top = radius.max()
    return radius / top if top > 0 else radius


def colorize(white: torch.Tensor, exponent: float, energy: float | None) -> torch.Tensor:
    "white noise shaped by clamp(f, eps)^(-exponent/2) over its non-unit axes, rescaled to the white std (or `energy`)"
    white_std = white.std()
    if exponent == 0.0:
        return white if energy is None or white_std < 1e-8 else white * (energy / white_std)
    body = white.squeeze()
    if body.dtype not in (torch.float32, torch.float64):
        body = body.to(torch.float32)
    spectrum = torch.fft.rfftn(body)  # (first, as in noise.py:377: a unit whose axes are all 1 long is refused by torch, with torch's error)
    mean_side = sum(body.shape) / len(body.shape) if body.shape else 1.0
    floor = 0.5 / max(mean_side, 4.0)  # half a bin: DC would diverge
    gain = torch.clamp(_radial_frequencies(tuple(body.shape), body.device), min=floor) ** (-exponent / 2.0)
    shaped = torch.fft.irfftn(spectrum * gain, s=body.shape)
    shaped_std = shaped.std()
    if shaped_std > 1e-8:
        shaped *= (white_std if energy is None else energy) / shaped_std
    return shaped.view(white.shape).to(dtype=white.dtype)


# ---- the batch ------------------------------------------------------------------------------------------------------
class HostStructuredBatch:
    """Offset / Pyramid / Colored for host-resident samples: one generator per batch item, results stacked -- the shape
    of the reference's BatchTensorNoise (noise.py:438-466) with the caller's own CPU generators."""

    def __init__(self, kind: type, unit_shape, seeds: list, props, dtype: torch.dtype = torch.float32):
        if kind is device_noise.Brownian:
            raise SkrampleHipError("Brownian noise on host tensors needs torchsde (the reference's own dependency), which this package does not stand in for on the CPU")
        if kind not in (device_noise.Offset, device_noise.Pyramid, device_noise.Colored):
            raise SkrampleHipError(f"no host evaluation of {getattr(kind, '__name__', kind)} noise")
        self.kind, self.unit, self.dtype = kind, tuple(unit_shape), dtype
        self.props = props if props is not None else {device_noise.Offset: OffsetProps, device_noise.Pyramid: PyramidProps, device_noise.Colored: ColoredProps}[kind]()
        self.generators = [s if isinstance(s, torch.Generator) else torch.Generator().manual_seed(device_noise.seed_value(s) & 0x7FFFFFFFFFFFFFFF) for s in seeds]
        self._sources = [_Draws(g, dtype) for g in self.generators]
        self._draws = 0
        # static variants freeze their added component when the generator is built (reference :98-102, :125-129)
        self._frozen: list[torch.Tensor] | None = None
        if getattr(self.props, "static", False):
            part = offset_component if kind is device_noise.Offset else pyramid_component
            self._frozen = [part(src, self.unit, self.props) for src in self._sources]

    def _one(self, i: int, step: Step | None) -> torch.Tensor:
        src = self._sources[i]
        if self.kind is device_noise.Offset:
            added = self._frozen[i] if self._frozen is not None else offset_component(src, self.unit, self.props)
            return src.normal(self.unit) + added
        if self.kind is device_noise.Pyramid:
            base = src.normal(self.unit)
            total = base + (self._frozen[i] if self._frozen is not None else pyramid_component(src, self.unit, self.props))
            return total / total.std()
        return colorize(src.normal(self.unit), colored_exponent(step, self.props), self.props.energy)

    def generate(self, step: Step | None = None) -> torch.Tensor:
        self._draws += 1
        return torch.stack([self._one(i, step) for i in range(len(self._sources))])

    generate_lazy = generate
