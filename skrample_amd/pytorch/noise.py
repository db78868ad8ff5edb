"""Device noise generators behind skrample's noise protocol.

Class names, constructor fields and `generate(step)` follow reference `skrample/pytorch/noise.py`
(SkrampleTensorNoise :17-24, TensorNoiseCommon :27-55, Random :58-74, Offset :77-113,
Pyramid :116-207, Colored :255-435, BatchTensorNoise :438-466).  What differs:

* Randomness is counter-based Philox4x32-10 keyed by a 64-bit per-sample seed (taken from
  `torch.Generator.initial_seed()` or an int), never a torch generator stream.  Draw `n` of a
  generator uses Philox stream id n*256 + k (k numbers the independent normals one draw needs), so
  any draw can be re-created later from (seed, stream) alone, results do not depend on batch
  sharding, and plain `Random` noise never has to exist in memory: `generate_lazy()` returns a
  symbolic `PhiloxNoise` that the step kernel draws in registers.
* A whole batch is one launch (`BatchTensorNoise` holds a device seed vector), not a Python loop
  over per-sample generators followed by `torch.stack`.
* Bit-level agreement with torch's CPU mt19937 / CUDA Philox streams is impossible by construction;
  the deterministic stages (offset broadcast, pyramid up-sampling, spectral colouring, per-sample
  normalisation) are parity-tested on injected draws (tests/test_noise_gpu.py).

`Brownian` (torchsde.BrownianInterval in the reference, :210-242) is a stateless virtual Brownian tree here:
W(t) is a fixed linear function of Philox-keyed node normals, walked on the host in fp64 and summed in one launch.
"""

from __future__ import annotations

import ctypes
import math
import os
import threading
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from functools import lru_cache
from typing import Any, Sequence

import torch

from .. import _hip
from .._hip import SkrampleHipError
from ..common import Step, divf, rescale_positive
from ..sampling.lazy import PhiloxNoise

SUBSTREAMS = 256  # Philox stream ids per draw


@dataclass(frozen=True)
class TensorNoiseProps:
    "configuration of a generator; reuse this, not the (stateful) generator"


def seed_value(seed) -> int:
    "64-bit key from a torch.Generator (its initial seed) or an int"
    if hasattr(seed, "initial_seed"):
        return seed.initial_seed() & 0xFFFFFFFFFFFFFFFF
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def seed_device(seed) -> torch.device:
    if isinstance(seed, torch.Generator) and seed.device.type == "cuda":
        return seed.device
    if not torch.cuda.is_available():
        raise SkrampleHipError("no HIP device available: skrample_amd noise generators run on the GPU only")
    return torch.device("cuda", torch.cuda.current_device())


_seed_vectors: dict = {}  # (device, seed values) -> device int64 vector: read-only by convention, so runs with the same seeds share one upload


class _PrivateVectors(threading.local):
    "per THREAD: > 0 means seed vectors are neither looked up nor kept (skrample_amd.graphs captures loops that overwrite theirs in place)"

    depth = 0

    def __getitem__(self, _i):  # (kept subscriptable: graphs.py counts it up and down as `_private_vectors[0] += 1`)
        return self.depth

    def __setitem__(self, _i, value):
        self.depth = value


_private_vectors = _PrivateVectors()
_captured_vectors: list = []  # shared vectors a caller's stream capture picked up (kept for the life of the process: a few hundred bytes each)


def forget_seed_vector(vector: torch.Tensor | None) -> None:
    "a holder is about to overwrite `vector` in place (a captured loop replayed with new seeds): it must not be shared any longer"
    for key in [k for k, v in _seed_vectors.items() if v is vector]:
        del _seed_vectors[key]


def seeds_tensor(values: Sequence[int], device: torch.device) -> torch.Tensor:
    key = (device, tuple(values))
    private = _private_vectors[0] > 0
    hit = None if private else _seed_vectors.get(key)
    if hit is not None:
        if device.type == "cuda":
            if torch.cuda.is_current_stream_capturing():
                _captured_vectors.append(hit)  # a caller's own graph will read it at every replay: it must outlive the cache's interest in it
            else:  # shared across wrappers and streams: tell the allocator this stream reads it too, so that the block is not recycled
                hit.record_stream(torch.cuda.current_stream(device))  # under a kernel still reading it once every holder let go
        return hit
    signed = [v - (1 << 64) if v >= (1 << 63) else v for v in values]
    out = torch.tensor(signed, dtype=torch.int64, device=device)
    if device.type == "cuda" and not private and not torch.cuda.is_current_stream_capturing():
        if len(_seed_vectors) >= 8:
            _seed_vectors.pop(next(iter(_seed_vectors)))
        _seed_vectors[key] = out
    return out


@dataclass
class SkrampleTensorNoise(ABC):
    @abstractmethod
    def generate(self, step: Step | None) -> torch.Tensor:
        "next noise tensor; stateful (advances the draw counter)"


@dataclass
class TensorNoiseCommon(SkrampleTensorNoise):
    shape: tuple[int, ...]
    seed: Any  # torch.Generator | int
    dtype: torch.dtype
    props: Any

    def __post_init__(self) -> None:
        self._draws = 0
        self._seeds_cache = None
        self._setup()

    @property
    def _device(self) -> torch.device:
        return seed_device(self.seed)

    @property
    def _seeds(self) -> torch.Tensor:
        "device seed vector (length 1), created on first single-generator use"
        if self._seeds_cache is None:
            self._seeds_cache = seeds_tensor([seed_value(self.seed)], self._device)
        return self._seeds_cache

    def _setup(self) -> None: ...

    def _next_stream(self) -> int:
        n = self._draws
        self._draws += 1
        return n * SUBSTREAMS

    @classmethod
    @abstractmethod
    def from_inputs(cls, shape, seed, props=None, dtype: torch.dtype = torch.float32): ...

    # batched core: every generator implements the whole-batch form; a single generator is batch 1
    @classmethod
    @abstractmethod
    def _batch(cls, unit_shape, seeds: torch.Tensor, stream: int, step: Step | None, props, dtype, state: dict) -> torch.Tensor: ...

    @classmethod
    def _batch_lazy(cls, unit_shape, seeds, stream, step, props, dtype, state):
        "symbolic result when the generator is plain white noise, else the realised tensor"
        return cls._batch(unit_shape, seeds, stream, step, props, dtype, state)

    def generate(self, step: Step | None) -> torch.Tensor:
        if not hasattr(self, "_state"):
            self._state = {}
        return self._batch(tuple(self.shape), self._seeds, self._next_stream(), step, self.props, self.dtype, self._state)[0]


@dataclass
class Random(TensorNoiseCommon):
    "plain standard-normal noise"

    @classmethod
    def from_inputs(cls, shape, seed, props=None, dtype=torch.float32):
        return cls(tuple(shape), seed, dtype, props)

    @classmethod
    def _batch(cls, unit_shape, seeds, stream, step, props, dtype, state):
        return cls._batch_lazy(unit_shape, seeds, stream, step, props, dtype, state).realize(dtype)

    @classmethod
    def _batch_lazy(cls, unit_shape, seeds, stream, step, props, dtype, state):
        return PhiloxNoise(seeds, stream, (seeds.shape[0], *unit_shape), seeds.device)


@dataclass(frozen=True)
class OffsetProps(TensorNoiseProps):
    dims: tuple[int, ...] = (0,)
    strength: float = 0.2
    static: bool = False


def _launch_ctx(seeds: torch.Tensor):
    return _hip.load(), _hip.current_stream_ptr(seeds.device)


@dataclass
class Offset(TensorNoiseCommon):
    "white noise plus a random offset shared along the dimensions NOT listed in `props.dims`"

    @classmethod
    def from_inputs(cls, shape, seed, props=OffsetProps(), dtype=torch.float32):
        return cls(tuple(shape), seed, dtype, props)

    def offset(self) -> torch.Tensor:
        "just the offset component of the next draw: strength^2 * N over the kept dims (reference noise.py:104-106)"
        from ..sampling import lazy as _lazy

        nd = len(self.shape)
        kept = {d + nd if d < 0 else d for d in self.props.dims}
        reduced = tuple(s if i in kept else 1 for i, s in enumerate(self.shape))
        noise = PhiloxNoise(self._seeds, self._next_stream() + 1, (1, *reduced), self._device)
        return _lazy.settle(_lazy.lift(noise) * float(self.props.strength) ** 2, dtype=self.dtype)[0]

    @classmethod
    def _batch(cls, unit_shape, seeds, stream, step, props, dtype, state):
        import ctypes

        if props.static:  # the offset is drawn once (first call) and reused
            offset_stream = state.setdefault("offset_stream", stream + 1)
        else:
            offset_stream = stream + 1
        nd_full = len(unit_shape)
        kept_dims = {d + nd_full if d < 0 else d for d in props.dims}
        # neighbouring dimensions that are both kept or both broadcast index the offset tensor as one merged dimension
        # (row-major), so any rank collapses to its runs; the kernel handles four
        merged: list[list] = []
        for i, size in enumerate(unit_shape):
            keep = i in kept_dims
            if merged and merged[-1][1] == keep:
                merged[-1][0] *= size
            else:
                merged.append([size, keep])
        if len(merged) > 4:
            raise SkrampleHipError("Offset noise supports up to four alternating runs of kept / broadcast dimensions per sample")
        nd = len(merged)
        mask = sum(1 << i for i, (_, keep) in enumerate(merged) if keep)
        out = torch.empty((seeds.shape[0], *unit_shape), dtype=dtype, device=seeds.device)
        lib, hstream = _launch_ctx(seeds)
        shape_arr = (ctypes.c_int64 * nd)(*[m[0] for m in merged])
        _hip.check(
            lib.skr_noise_offset(out.data_ptr(), _hip.DTYPE_CODE[dtype], seeds.data_ptr(), stream, offset_stream, seeds.shape[0], shape_arr, nd, mask, float(props.strength), hstream),
            "skr_noise_offset",
        )
        return out


@dataclass(frozen=True)
class PyramidProps(OffsetProps):
    dims: tuple[int, ...] = (-1, -2)
    strength: float = 0.3
    depth: int = 99


@dataclass(frozen=True)
class BrownianProps(TensorNoiseProps):
    max_steps: int = 10_000


@dataclass(frozen=True)
class ColoredProps(TensorNoiseProps):
    energy: float | None = None
    color_start: float = 1 / 4
    color_end: float = -2
    color_curve: float = 2


def colored_exponent(step: Step | None, props: ColoredProps) -> float:
    "power-law exponent for this step: color_start at the beginning, color_end at time_to = 1"
    if step is None:
        return props.color_start
    if props.color_curve == math.inf:
        return props.color_end
    t = step.normal().clamp().time_to
    shift = rescale_positive(-props.color_curve)
    t = shift / (shift + (divf(1, t) - 1))
    return (1 - t) * props.color_start + t * props.color_end


PYRAMID_MAX_LEVELS = 8


def pyramid_level_tables(unit_hw: tuple[int, int], resize_h: bool, uniforms):
    """Per-sample level geometry, vectorised over the batch.  uniforms: [batch, PYRAMID_MAX_LEVELS] in [0,1).
    Returns (table int32 [batch][PYRAMID_MAX_LEVELS][2], counts int32 [batch]).  Level i shrinks the *running*
    size by r_i**i, r_i = 2 + 2*u_i, and the pyramid ends with the first level that has a resized dimension of
    1 (reference noise.py:157-162,195-196; its 99-level cap cannot be reached: sizes fall super-geometrically)."""
    import numpy as np

    batch = uniforms.shape[0]
    h = np.full(batch, unit_hw[0], dtype=np.int64)
    w = np.full(batch, unit_hw[1], dtype=np.int64)
    table = np.zeros((batch, PYRAMID_MAX_LEVELS, 2), dtype=np.int32)
    counts = np.zeros(batch, dtype=np.int32)
    alive = np.ones(batch, dtype=bool)
    for i in range(PYRAMID_MAX_LEVELS):
        shrink = (uniforms[:, i] * 2 + 2) ** i
        if resize_h:
            h = np.where(alive, np.maximum(1, (h / shrink).astype(np.int64)), h)
        w = np.where(alive, np.maximum(1, (w / shrink).astype(np.int64)), w)
        table[alive, i, 0] = h[alive]
        table[alive, i, 1] = w[alive]
        counts[alive] = i + 1
        alive &= ~((w <= 1) | (resize_h & (h <= 1)))
        if not alive.any():
            return table, counts
    raise SkrampleHipError("pyramid deeper than PYRAMID_MAX_LEVELS levels (per-sample sizes this large are not supported)")


@dataclass
class Pyramid(TensorNoiseCommon):
    """Multi-resolution noise (white noise + bilinearly up-sampled coarser noise levels), rescaled to unit
    variance per sample.  Level geometry is random per sample, as in the reference."""

    @classmethod
    def from_inputs(cls, shape, seed, props=PyramidProps(), dtype=torch.float32):
        return cls(tuple(shape), seed, dtype, props)

    @staticmethod
    def _geometry(unit_shape, props) -> tuple[int, int, int, bool]:
        "(lead, h, w, resize_h) when the resized axes are the trailing one or two (the fast LDS kernels); else None"
        nd = len(unit_shape)
        dims = sorted({(d + nd if d < 0 else d) for d in props.dims})
        if dims == [nd - 2, nd - 1] and nd >= 2:
            return math.prod(unit_shape[:-2]), unit_shape[-2], unit_shape[-1], True
        if dims == [nd - 1]:
            return math.prod(unit_shape[:-1]), 1, unit_shape[-1], False
        return None

    @staticmethod
    def _axes_nd(unit_shape, props) -> tuple[list[int], int, int]:
        """Any pair of resized axes (reference noise.py:146-193: resized axes are permuted to the end, interpolated slice by
        slice and permuted back; level tensors are drawn in the unit's own axis order).  Adjacent kept axes are merged so
        that the kernel sees at most four: (shape, axis_a, axis_b)."""
        nd = len(unit_shape)
        dims = sorted({(d + nd if d < 0 else d) for d in props.dims})
        if any(not 0 <= d < nd for d in dims):
            raise SkrampleHipError(f"Pyramid dims {tuple(props.dims)} outside a {nd}-axis unit")
        if len(dims) == 1:
            # (the trailing axis alone is the linear case handled by _geometry)
            raise SkrampleHipError("Pyramid over a single non-trailing axis: the reference itself fails there (RuntimeError in its permute, noise.py:176); resize the last axis or two axes")
        if len(dims) != 2:
            raise SkrampleHipError("Pyramid resizes one or two axes (three would be the reference's 'bicubic' on 5-D input, which torch.interpolate rejects)")
        shape, axes, run = [], [], None
        for ax, size in enumerate(unit_shape):
            if ax in dims:
                axes.append(len(shape))
                shape.append(size)
                run = None
            elif run is None:
                run = len(shape)
                shape.append(size)
            else:
                shape[run] *= size
        if len(shape) > 4:
            raise SkrampleHipError(f"Pyramid dims {tuple(props.dims)} split the unit {tuple(unit_shape)} into more than four axis groups")
        return shape, axes[0], axes[1]

    def pyramid(self) -> torch.Tensor:
        "just the added 'pyramid' component of the next draw, un-normalised (reference noise.py:146-200)"
        if not hasattr(self, "_state"):
            self._state = {}
        return self._batch(tuple(self.shape), self._seeds, self._next_stream(), None, self.props, self.dtype, self._state, with_base=False)[0]

    @classmethod
    def _batch(cls, unit_shape, seeds, stream, step, props, dtype, state, with_base: bool = True):
        # static: the pyramid component is frozen at the first draw (same streams every time), only the base changes
        stream_levels = state.setdefault("static_stream", stream) if props.static else stream
        geometry = cls._geometry(unit_shape, props)
        if geometry is None:
            return cls._batch_nd(unit_shape, seeds, stream, stream_levels, props, dtype, state, with_base)
        lead, h, w, resize_h = geometry
        batch = seeds.shape[0]
        dev = seeds.device
        key = ("ws", batch, lead, h, w)
        if key not in state:
            for k in [k for k in state if k != "static_stream"]:
                del state[k]  # (also drops the any-shape workspaces of a previous shape)
            state[key] = (
                torch.empty(batch * lead * h * w, dtype=torch.float32, device=dev),
                torch.empty(batch * lead * 2, dtype=torch.float64, device=dev),
                torch.empty(batch * (PYRAMID_MAX_LEVELS * 2 + 1), dtype=torch.int32, device=dev),
            )
        scratch, partials, levels = state[key]
        out = torch.empty((batch, *unit_shape), dtype=dtype, device=dev)
        lib, hstream = _launch_ctx(seeds)
        tail = (seeds.data_ptr(), stream, stream_levels, batch, lead, h, w, 1 if resize_h else 0, float(props.strength), int(min(props.depth, 1 << 20)), 1 if with_base else 0, hstream)
        status = _hip.SKR_ERR_UNSUPPORTED if state.get("any_shape") else lib.skr_noise_pyramid(out.data_ptr(), _hip.DTYPE_CODE[dtype], scratch.data_ptr(), partials.data_ptr(), levels.data_ptr(), *tail)
        if status == _hip.SKR_ERR_UNSUPPORTED:
            # planes too large for the LDS level stage, or a width that is not a multiple of 4: levels in global memory
            if "any_shape" not in state:
                slots = max(1, min(1024, -(-(lead * h * w) // (4 * 256 * 8))))
                state["any_shape"] = (torch.empty(batch * lead * h * w, dtype=torch.float32, device=dev), torch.empty(batch * slots * 2, dtype=torch.float64, device=dev), slots)
            level_ws, partials_any, slots = state["any_shape"]
            status = lib.skr_noise_pyramid_any(out.data_ptr(), _hip.DTYPE_CODE[dtype], scratch.data_ptr(), level_ws.data_ptr(), partials_any.data_ptr(), slots, levels.data_ptr(), *tail)
            _hip.check(status, "skr_noise_pyramid_any")
        else:
            _hip.check(status, "skr_noise_pyramid")
        state["levels"] = levels  # device table of the last draw: [batch][8][2] sizes, then [batch] counts
        return out

    @classmethod
    def _batch_nd(cls, unit_shape, seeds, stream, stream_levels, props, dtype, state, with_base: bool):
        "resized axes anywhere in the unit: the any-shape kernels with an axis descriptor (skr_noise_pyramid_nd)"
        shape, axis_a, axis_b = cls._axes_nd(unit_shape, props)
        batch, dev, unit = seeds.shape[0], seeds.device, math.prod(unit_shape)
        key = ("ws_nd", batch, tuple(shape), axis_a, axis_b)
        if key not in state:
            for k in [k for k in state if k != "static_stream"]:
                del state[k]
            slots = max(1, min(1024, -(-unit // (4 * 256 * 8))))
            state[key] = (
                torch.empty(batch * unit, dtype=torch.float32, device=dev),  # scratch
                torch.empty(batch * unit, dtype=torch.float32, device=dev),  # level normals
                torch.empty(batch * slots * 2, dtype=torch.float64, device=dev),
                torch.empty(batch * (PYRAMID_MAX_LEVELS * 2 + 1), dtype=torch.int32, device=dev),
                slots,
            )
        scratch, level_ws, partials, levels, slots = state[key]
        out = torch.empty((batch, *unit_shape), dtype=dtype, device=dev)
        lib, hstream = _launch_ctx(seeds)
        status = lib.skr_noise_pyramid_nd(
            out.data_ptr(), _hip.DTYPE_CODE[dtype], scratch.data_ptr(), level_ws.data_ptr(), partials.data_ptr(), slots, levels.data_ptr(), seeds.data_ptr(),
            stream, stream_levels, batch, len(shape), (ctypes.c_int64 * len(shape))(*shape), axis_a, axis_b, float(props.strength), int(min(props.depth, 1 << 20)),
            1 if with_base else 0, hstream,
        )  # fmt: skip
        _hip.check(status, "skr_noise_pyramid_nd")
        state["levels"] = levels
        return out


@dataclass
class Colored(TensorNoiseCommon):
    """Power-law coloured noise: white Philox noise shaped in the Fourier domain by f^(-exponent/2), the
    exponent moving from `color_start` to `color_end` over the schedule; per-sample std preserved (or set to
    `energy`).  Power-of-two shapes (every BASELINE config) and planes with sides 2^a x (odd <= 63) run on the hand-written LDS
    plane kernels; any other shape takes the same pipeline axis by axis on the library's own any-length transforms
    (csrc/skr_fft_own.hip: Bluestein on the same LDS tile transform; axes up to 2048 long, powers of two up to 4096)."""

    @classmethod
    def from_inputs(cls, shape, seed, props=ColoredProps(), dtype=torch.float32):
        return cls(tuple(shape), seed, dtype, props)

    @staticmethod
    def _axes(unit_shape) -> tuple[list[int], bool]:
        "(transform dims after dropping size-1 axes, whether the hand-written power-of-two path applies)"
        dims = [d for d in unit_shape if d != 1]
        if not 1 <= len(dims) <= 12 or any(d > 128 for d in dims[:-3]):
            raise SkrampleHipError(f"Colored noise needs 1 to 12 transform axes per sample (those outside the last three at most 128 long), got shape {tuple(unit_shape)}")
        pow2 = 2 <= len(dims) <= 3 and all(d & (d - 1) == 0 for d in dims) and dims[-1] >= 4 and max(dims) <= 4096
        return dims, pow2

    @staticmethod
    def _mixed_radix_candidate(dims: list[int]) -> bool:
        "2-D / 3-D units the mixed-radix plane kernel may take (it decides itself: sides 2^a * odd factor <= 63, plane fits LDS)"
        if not 2 <= len(dims) <= 3 or dims[-1] % 4 or dims[-2] % 2 or max(dims) > 4096:
            return False
        return len(dims) == 2 or (dims[0] <= 16 and dims[0] & (dims[0] - 1) == 0)

    @staticmethod
    def colorize_noise(white: torch.Tensor, exponent: float = 0.0, energy: float | None = None) -> torch.Tensor:
        """Colour an existing white-noise tensor with the power-law spectrum f^(-exponent), normalised back to the
        input's std (or to `energy`).  Size-1 dimensions are excluded from the transform; no batching -- the whole
        tensor is one sample (reference noise.py:337-403).  Any shape with 1-6 transform axes (the library's own
        any-length transforms on the inner three + direct outer-axis DFTs)."""
        import ctypes

        _hip.require_device(white, "white noise")
        if exponent == 0.0:
            if energy is None:
                return white
            wstd = white.float().std()
            return white if wstd.item() < 1e-8 else (white.float() * (energy / wstd)).to(white.dtype)
        dims = [d for d in white.shape if d != 1]
        if not 1 <= len(dims) <= 6 or any(d > 128 for d in dims[:-3]):
            raise SkrampleHipError(f"colorize_noise needs 1 to 6 transform axes (those outside the last three at most 128 long), got shape {tuple(white.shape)}")
        dev, unit = white.device, math.prod(dims)
        work = white.detach().to(torch.float32).contiguous().clone().reshape(-1)  # transform workspace, overwritten
        spec = torch.empty(unit // dims[-1] * (dims[-1] // 2 + 1), dtype=torch.complex64, device=dev)
        partials = torch.empty(4 * 256, dtype=torch.float64, device=dev)
        out_dtype = white.dtype if white.dtype in _hip.DTYPE_CODE else torch.float32
        out = torch.empty(white.shape, dtype=out_dtype, device=dev)
        status = _hip.load().skr_colorize(
            out.data_ptr(), _hip.DTYPE_CODE[out_dtype], spec.data_ptr(), work.data_ptr(), partials.data_ptr(), 1, len(dims), (ctypes.c_int32 * len(dims))(*dims),
            float(exponent), 0 if energy is None else 1, 0.0 if energy is None else float(energy), _hip.current_stream_ptr(dev),
        )
        _hip.check(status, "skr_colorize")
        return out

    @classmethod
    def _batch_lazy(cls, unit_shape, seeds, stream, step, props, dtype, state):
        if colored_exponent(step, props) == 0.0 and props.energy is None:
            return PhiloxNoise(seeds, stream, (seeds.shape[0], *unit_shape), seeds.device)  # plain white noise
        return cls._batch(unit_shape, seeds, stream, step, props, dtype, state)

    @classmethod
    def _batch(cls, unit_shape, seeds, stream, step, props, dtype, state):
        import ctypes

        exponent = colored_exponent(step, props)
        batch = seeds.shape[0]
        if exponent == 0.0 and props.energy is None:
            return PhiloxNoise(seeds, stream, (batch, *unit_shape), seeds.device).realize(dtype)
        dims, pow2 = cls._axes(unit_shape)
        dev = seeds.device
        unit = math.prod(dims)
        half = unit // dims[-1] * (dims[-1] // 2 + 1)
        key = ("ws", batch, tuple(dims))
        if key not in state:
            slots = max(256, -(-(unit // dims[-1]) // max(2, 2 * (256 // dims[-1]))))
            state.clear()
            state[key] = (
                torch.empty(batch * half, dtype=torch.complex64, device=dev),
                torch.empty(batch * unit, dtype=torch.float32, device=dev),
                torch.empty(4 * batch * slots, dtype=torch.float64, device=dev),
                slots,
            )
        spec, scratch, partials, slots = state[key]
        out = torch.empty((batch, *unit_shape), dtype=dtype, device=dev)
        lib, hstream = _launch_ctx(seeds)
        has_energy, energy = (0, 0.0) if props.energy is None else (1, float(props.energy))
        status = _hip.SKR_ERR_UNSUPPORTED
        if pow2 or cls._mixed_radix_candidate(dims):
            # hand-written LDS transforms: powers of two, and planes with one factor 3 or 5 per side (96, 160 ... under a
            # power-of-two channel axis); the library answers SKR_ERR_UNSUPPORTED for what its kernels do not cover
            d1, d2, d3 = ([1] + dims)[-3:]
            status = lib.skr_noise_colored(
                out.data_ptr(), _hip.DTYPE_CODE[dtype], spec.data_ptr(), scratch.data_ptr(), partials.data_ptr(), slots, seeds.data_ptr(), stream,
                batch, d1, d2, d3, float(exponent), has_energy, energy, hstream,
            )
            if pow2 or status != _hip.SKR_ERR_UNSUPPORTED:
                _hip.check(status, "skr_noise_colored")
        if status == _hip.SKR_ERR_UNSUPPORTED:
            status = lib.skr_noise_colored_any(
                out.data_ptr(), _hip.DTYPE_CODE[dtype], spec.data_ptr(), scratch.data_ptr(), partials.data_ptr(), seeds.data_ptr(), stream,
                batch, len(dims), (ctypes.c_int32 * len(dims))(*dims), float(exponent), has_energy, energy, hstream,
            )
            _hip.check(status, "skr_noise_colored_any")
        return out


BROWNIAN_STREAMS = 1 << 63  # Philox stream namespace of the tree nodes (disjoint from the per-draw streams n*256+k)


def brownian_depth(max_steps: int) -> int:
    "dyadic levels below [0,1]: leaves are 4x finer than the reference's tolerance 1/(10*max_steps) (noise.py:230)"
    return math.ceil(math.log2(max_steps * 10)) + 2


def brownian_path(t: float, depth: int) -> dict[int, float]:
    """W(t), t in [0,1], as weights over independent standard normals.  Node 0 is W(1); node h >= 1 is the
    Brownian-bridge normal of the dyadic interval with heap index h (root [0,1] = 1, children 2h / 2h+1):
    W(mid) = (W(a) + W(b))/2 + sqrt(b-a)/2 * Z_h.  Below `depth` levels the path is linear inside the leaf.
    All arithmetic is on dyadic rationals, exact in fp64 except the final interpolation."""
    lo, hi, node = 0.0, 1.0, 1
    w_lo: dict[int, float] = {}
    w_hi: dict[int, float] = {0: 1.0}
    for _ in range(depth):
        if t == lo or t == hi:
            break
        mid = 0.5 * (lo + hi)
        w_mid = {k: 0.5 * (w_lo.get(k, 0.0) + w_hi.get(k, 0.0)) for k in {*w_lo, *w_hi}}
        w_mid[node] = 0.5 * math.sqrt(hi - lo)
        if t < mid:
            hi, w_hi, node = mid, w_mid, 2 * node
        else:
            lo, w_lo, node = mid, w_mid, 2 * node + 1
    f = (t - lo) / (hi - lo)
    return {k: (1 - f) * w_lo.get(k, 0.0) + f * w_hi.get(k, 0.0) for k in {*w_lo, *w_hi}}


# ---- the schedule's own partition (round 4) ----------------------------------------------------------------------------
# A sampling run asks for Step.from_int(k, N): both endpoints are points of the N-cell partition of [0, 1], never dyadic for the usual
# N (20, 30, 50), so the dyadic tree above walks all `depth` = 19 levels for each of them -- 20 normals per element and endpoint.
# A generator whose FIRST query is one cell of such a partition therefore fixes N for its lifetime and builds its path on the
# partition instead: W at the grid points by bisection of the INDEX range (W(1) = Z_0, then for the index interval (lo, hi) with
# heap index h and mid = (lo + hi) // 2 the bridge  W(mid/N) = ((hi-mid) W(lo/N) + (mid-lo) W(hi/N)) / (hi-lo)
# + sqrt((mid-lo)(hi-mid) / ((hi-lo) N)) Z_h ), ceil(log2 N) levels, and W inside a cell by the dyadic bridge between the cell's two
# grid values (the levels that are left of `depth`).  Grid queries -- every step of a run, RK stages included -- cost 5-6 normals per
# endpoint instead of 20; any other query stays consistent with them because it is evaluated on the SAME path.  What changes against
# the pure dyadic tree: the path of a seed now depends on the partition the generator was first asked about (a generator first asked
# an off-partition step keeps the dyadic tree).  torchsde's values cannot be reproduced either way (module header).
BROWNIAN_GRID_NODE = 1 << 61  # stream namespace of index-bisection nodes (| heap index)
BROWNIAN_CELL_NODE = 1 << 60  # stream namespace of in-cell bridge nodes (| cell << 24 | heap index inside the cell)
BROWNIAN_MAX_GRID = 1 << 20
BROWNIAN_SCHEDULE_PARTITION = os.environ.get("SKR_BROWNIAN_PARTITION", "") not in ("", "0")  # see Brownian._batch


def brownian_grid_index(t: float, n: int) -> int | None:
    "j with j / n == t exactly (the float Step.from_int produces), else None"
    j = round(t * n)
    return j if 0 <= j <= n and j / n == t else None


def brownian_grid_of(time_from: float, time_to: float) -> int | None:
    "N if (time_from, time_to) is exactly one cell of the N-cell partition of [0, 1] (what Step.from_int(k, N) yields), else None"
    width = time_to - time_from
    if not width > 0:
        return None
    n = round(1.0 / width)
    if not 1 <= n <= BROWNIAN_MAX_GRID:
        return None
    j = brownian_grid_index(time_from, n)
    return n if j is not None and brownian_grid_index(time_to, n) == j + 1 else None


def brownian_grid_point(j: int, n: int) -> dict[int, float]:
    "W(j / n) as weights over the terminal normal (key 0) and the index-bisection normals (keys BROWNIAN_GRID_NODE | heap index)"
    lo, hi, node = 0, n, 1
    w_lo: dict[int, float] = {}
    w_hi: dict[int, float] = {0: 1.0}
    while True:
        if j == lo:
            return dict(w_lo) if w_lo else {0: 0.0}
        if j == hi:
            return dict(w_hi)
        mid = (lo + hi) // 2
        a, b = (hi - mid) / (hi - lo), (mid - lo) / (hi - lo)
        w_mid = {k: a * w_lo.get(k, 0.0) + b * w_hi.get(k, 0.0) for k in {*w_lo, *w_hi}}
        w_mid[BROWNIAN_GRID_NODE | node] = math.sqrt((mid - lo) * (hi - mid) / ((hi - lo) * n))
        if j < mid:
            hi, w_hi, node = mid, w_mid, 2 * node
        else:
            lo, w_lo, node = mid, w_mid, 2 * node + 1


def brownian_grid_path(t: float, n: int, depth: int) -> dict[int, float]:
    "W(t) on the path built over the n-cell partition: a grid point, or the dyadic bridge inside its cell"
    j = brownian_grid_index(t, n)
    if j is not None:
        return brownian_grid_point(j, n)
    cell = min(int(t * n), n - 1)
    if not cell / n < t < (cell + 1) / n:  # (rounding of t * n at a cell boundary)
        cell = cell - 1 if t < cell / n else cell + 1
    lo, hi, node = cell / n, (cell + 1) / n, 1
    w_lo, w_hi = brownian_grid_point(cell, n), brownian_grid_point(cell + 1, n)
    for _ in range(max(depth - (n - 1).bit_length(), 2)):
        if t == lo or t == hi:
            break
        mid = 0.5 * (lo + hi)
        w_mid = {k: 0.5 * (w_lo.get(k, 0.0) + w_hi.get(k, 0.0)) for k in {*w_lo, *w_hi}}
        w_mid[BROWNIAN_CELL_NODE | (cell << 24) | node] = 0.5 * math.sqrt(hi - lo)
        if t < mid:
            hi, w_hi, node = mid, w_mid, 2 * node
        else:
            lo, w_lo, node = mid, w_mid, 2 * node + 1
    f = (t - lo) / (hi - lo)
    return {k: (1 - f) * w_lo.get(k, 0.0) + f * w_hi.get(k, 0.0) for k in {*w_lo, *w_hi}}


@lru_cache(maxsize=4096)
def brownian_increment(time_from: float, time_to: float, depth: int) -> tuple[tuple[int, ...], tuple[float, ...]]:
    "(node ids, weights) of (W(time_to) - W(time_from)) / sqrt(time_to - time_from); unit variance up to the leaf interpolation"
    a, b = brownian_path(time_from, depth), brownian_path(time_to, depth)
    scale = 1.0 / math.sqrt(time_to - time_from)
    nodes = sorted({*a, *b})
    weights = [(b.get(k, 0.0) - a.get(k, 0.0)) * scale for k in nodes]
    keep = [(k, w) for k, w in zip(nodes, weights) if w != 0.0]
    return tuple(k for k, _ in keep), tuple(w for _, w in keep)


@lru_cache(maxsize=4096)
def brownian_endpoints(time_from: float | None, time_to: float, depth: int, grid: int | None = None) -> tuple[tuple[int, ...], tuple[float, ...], tuple[float, ...]]:
    """(ascending node ids, weights of W(time_to), weights of W(time_from)) over the union of both paths; with
    time_from = None only W(time_to)'s own nodes (the caller has W(time_from) cached).  `grid` = N: the path built over the
    N-cell partition (brownian_grid_path) instead of the dyadic tree."""
    path = (lambda t: brownian_grid_path(t, grid, depth)) if grid else (lambda t: brownian_path(t, depth))
    b = path(time_to)
    a = path(time_from) if time_from is not None else {}
    nodes = tuple(sorted({*a, *b}))
    return nodes, tuple(b.get(k, 0.0) for k in nodes), tuple(a.get(k, 0.0) for k in nodes)


@dataclass
class Brownian(TensorNoiseCommon):
    """Noise that is a deterministic function of the Step: increments of one fixed Brownian path per seed, so
    overlapping / adjacent steps are consistent (reference noise.py:210-242, there via torchsde.BrownianInterval).
    Here the path is a virtual dyadic tree keyed by Philox -- every query is one launch and needs no tree state.
    The only state is an optimisation: W(time_to) of the last query is kept (fp32), so a query that starts where
    the previous one ended -- every step of a sampling loop -- evaluates one path instead of two, with identical bits."""

    @classmethod
    def from_inputs(cls, shape, seed, props=BrownianProps(), dtype=torch.float32):
        return cls(tuple(shape), seed, dtype, props)

    @classmethod
    def _batch(cls, unit_shape, seeds, stream, step, props, dtype, state):
        import ctypes

        if not step:
            return Random._batch(unit_shape, seeds, stream, step, props, dtype, state)
        step = step.normal().clamp()
        depth = brownian_depth(props.max_steps)
        t0, t1 = float(step.time_from), float(step.time_to)
        shape = (seeds.shape[0], *unit_shape)
        cache = state.get("brownian_cache")
        if cache is None or tuple(cache.shape) != shape or cache.device != seeds.device:
            cache = state["brownian_cache"] = torch.empty(shape, dtype=torch.float32, device=seeds.device)
            state["brownian_cache_time"] = None
        if "brownian_grid" not in state:
            # Default: the dyadic tree -- W(t) is a function of (seed, t) alone, so the same seed walks the same path whatever the step count
            # or the order of the queries, as torchsde's BrownianInterval does for the reference.  Opt-in
            # (BROWNIAN_SCHEDULE_PARTITION / SKR_BROWNIAN_PARTITION=1): the path is built over the N-cell partition the FIRST query
            # belongs to -- 6x cheaper per step (0.21 against 1.33 ms at 256 x (16,128,128)), but a 20-step and a 30-step run of one seed
            # then see different paths.
            state["brownian_grid"] = brownian_grid_of(t0, t1) if BROWNIAN_SCHEDULE_PARTITION else None
        hit = state.get("brownian_cache_time") == t0
        nodes, w_to, w_from = brownian_endpoints(None if hit else t0, t1, depth, state["brownian_grid"])
        if len(nodes) > 64:
            raise SkrampleHipError(f"Brownian max_steps={props.max_steps} needs {len(nodes)} tree nodes per query (limit 64)")
        out = torch.empty(shape, dtype=dtype, device=seeds.device)
        lib, hstream = _launch_ctx(seeds)
        ids = (ctypes.c_uint64 * len(nodes))(*[BROWNIAN_STREAMS | n for n in nodes])
        wt = (ctypes.c_double * len(nodes))(*w_to)
        wf = (ctypes.c_double * len(nodes))(*w_from)
        status = lib.skr_noise_brownian(out.data_ptr(), _hip.DTYPE_CODE[dtype], seeds.data_ptr(), ids, wt, wf, len(nodes), 1.0 / math.sqrt(t1 - t0),
                                        cache.data_ptr(), 1 if hit else 0, seeds.shape[0], math.prod(unit_shape), hstream)
        _hip.check(status, "skr_noise_brownian")
        state["brownian_cache_time"] = t1
        return out




class HostRandomBatch:
    """White noise for host-resident samples (CPU tensors), exactly as the reference draws it: one torch generator per
    batch item, `torch.randn(unit_shape, generator=g)` each, stacked (noise.py:58-74, 438-446; fp32 because the
    generators live on the CPU, diffusers.py:343).  Structured generators (Offset / Pyramid / Colored / Brownian) are HIP
    kernels and exist on the device only."""

    def __init__(self, unit_shape, seeds: list, dtype: torch.dtype = torch.float32):
        self.unit_shape = tuple(unit_shape)
        self.dtype = dtype
        self.generators = [s if isinstance(s, torch.Generator) else torch.Generator().manual_seed(seed_value(s) & 0x7FFFFFFFFFFFFFFF) for s in seeds]
        self._draws = 0

    def generate(self, step=None) -> torch.Tensor:
        self._draws += 1
        return torch.stack([torch.randn(self.unit_shape, generator=g, dtype=self.dtype, device=g.device) for g in self.generators])

    generate_lazy = generate


class _LazyGenerators(Sequence):
    """The per-item generators of a batch, built on first access.  The reference builds one generator object per batch item
    (noise.py:438-446) and so does `BatchTensorNoise.generators`; here the batch runs on its seed vector as ONE launch, and 256
    dataclass constructions per run (0.25-0.3 ms of host time at B = 256, at the first step of every run) are paid only by code that
    actually looks at the list."""

    def __init__(self, subclass, unit_shape, seeds, props, dtype):
        self._subclass, self._unit_shape, self.raw_seeds, self._props, self._dtype = subclass, unit_shape, list(seeds), props, dtype
        self._made: dict[int, TensorNoiseCommon] = {}

    def _make(self, i: int):
        got = self._made.get(i)
        if got is None:
            s = self.raw_seeds[i]
            got = self._subclass.from_inputs(self._unit_shape, s, self._props, self._dtype) if self._props is not None else self._subclass.from_inputs(self._unit_shape, s, dtype=self._dtype)
            self._made[i] = got
        return got

    def __len__(self) -> int:
        return len(self.raw_seeds)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._make(j) for j in range(*i.indices(len(self)))]
        n = len(self)
        if not -n <= i < n:
            raise IndexError(i)
        return self._make(i % n)

    def __eq__(self, other) -> bool:
        return isinstance(other, Sequence) and len(other) == len(self) and all(a == b for a, b in zip(self, other))

    def __repr__(self) -> str:
        return f"<{len(self)} x {self._subclass.__name__} generators, built on access>"


@dataclass
class BatchTensorNoise(SkrampleTensorNoise):
    """One logical generator per batch item, executed as a single launch.  `generators` is kept for
    API compatibility (len == batch); the batch shares one draw counter."""

    generators: list[TensorNoiseCommon]
    _draws: int = field(default=0, repr=False)

    def __post_init__(self) -> None:
        if not self.generators:
            raise ValueError("BatchTensorNoise needs at least one generator")
        first = self.generators[0]
        self._kind = type(first)
        self._device = first._device
        raw = self.generators.raw_seeds if isinstance(self.generators, _LazyGenerators) else [g.seed for g in self.generators]
        self._seeds = seeds_tensor([v & 0xFFFFFFFFFFFFFFFF if type(v) is int else seed_value(v) for v in raw], self._device)
        self._state: dict = {}
        self.seeds_ptr = self._seeds.data_ptr()
        self.batch_shape = (len(raw), *first.shape)  # shape of one draw

    def _stream(self) -> int:
        n = self._draws
        self._draws += 1
        return n * SUBSTREAMS

    def generate(self, step: Step | None) -> torch.Tensor:
        g = self.generators[0]
        return self._kind._batch(tuple(g.shape), self._seeds, self._stream(), step, g.props, g.dtype, self._state)

    def generate_lazy(self, step: Step | None):
        "PhiloxNoise (drawn inside the step kernel) for white noise, a realised tensor otherwise"
        g = self.generators[0]
        return self._kind._batch_lazy(tuple(g.shape), self._seeds, self._stream(), step, g.props, g.dtype, self._state)

    def workspace_tensors(self) -> list[torch.Tensor]:
        "every device buffer the generator keeps between draws (scratch, partial sums, level tables, caches) and its seed vector"
        found = [self._seeds]

        def walk(v) -> None:
            if isinstance(v, torch.Tensor):
                found.append(v)
            elif isinstance(v, (tuple, list)):
                for w in v:
                    walk(w)

        for v in self._state.values():
            walk(v)
        return found

    def used_on(self, stream: "torch.cuda.Stream") -> None:
        """The buffers above were just used by launches on `stream`.  They belong to whichever stream was current when they were
        allocated; telling the allocator about the other one keeps a buffer that is dropped later (a shape change re-allocates
        the workspaces) from being handed out again while that stream may still be writing it."""
        for t in self.workspace_tensors():
            if t.is_cuda:
                t.record_stream(stream)

    @classmethod
    def from_batch_inputs(cls, subclass, unit_shape, seeds: list, props=None, dtype: torch.dtype = torch.float32) -> "BatchTensorNoise":
        return cls(_LazyGenerators(subclass, tuple(unit_shape), seeds, props, dtype))
