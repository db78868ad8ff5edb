"""Noise generators (torch CPU restatement) and the counter-based RNG specification used by the
HIP kernels.  Follows reference skrample/pytorch/noise.py.  Brownian is the exception: its arithmetic
lives in the un-vendored dependency torchsde (>=0.2.6, pyproject.toml:22; call site noise.py:222-242), so
`brownian_noise` below restates the published construction (Brownian-bridge bisection of [0,1]; Li et al. 2020
"Scalable gradients for SDEs" virtual Brownian tree / Kidger et al. 2021 Brownian Interval) on the Philox
streams and is PARITY UNPINNED against torchsde's own random values -- only the law (variance, independence,
additivity over adjacent steps) and the reference's call-site arithmetic (normalise, clamp, / sqrt(dt)) are pinned.

Every generator takes its random draws through small callables (`randn(shape)`, `rand1()`), so the
deterministic part (offset broadcast, pyramid up-sampling + blend, spectral colouring, per-sample
normalisation) can be compared with the HIP kernels on *injected* draws: bit-level RNG parity
between torch's CPU mt19937 stream and a GPU Philox stream is impossible by construction.

Second half: Philox4x32-10 (Salmon et al., SC'11 "Parallel random numbers: as easy as 1, 2, 3";
Random123 v1.14 `philox.h`) + Box-Muller, restated in numpy.  This is the *specification* of the
in-kernel RNG of skrample_amd/csrc: the kernels must reproduce `philox_normal` to float tolerance
and `philox4x32` bit for bit.
"""

from __future__ import annotations

import math
from typing import Callable

import numpy as np
import torch

from .scalars import divf, rescale_positive, stp_clamp, stp_normal


# ---------------------------------------------------------------------------------------------------
# torch-generator draw helpers (noise.py:36-42)
# ---------------------------------------------------------------------------------------------------
def torch_draws(gen: torch.Generator, dtype=torch.float32):
    def randn(shape):
        return torch.randn(tuple(shape), generator=gen, dtype=dtype, device=gen.device)

    def rand1() -> float:
        return torch.rand([1], dtype=dtype, device=gen.device, generator=gen).item()

    return randn, rand1


class Recorder:
    "wraps draw callables and remembers every draw (so fixtures can carry the consumed randoms)"

    def __init__(self, randn, rand1):
        self._randn, self._rand1 = randn, rand1
        self.normals: list[torch.Tensor] = []
        self.uniforms: list[float] = []

    def randn(self, shape):
        v = self._randn(shape)
        self.normals.append(v.clone())
        return v

    def rand1(self) -> float:
        v = self._rand1()
        self.uniforms.append(v)
        return v


class Replay:
    "feeds recorded draws back in order"

    def __init__(self, normals, uniforms=()):
        self.normals, self.uniforms = list(normals), list(uniforms)

    def randn(self, shape):
        v = self.normals.pop(0)
        assert tuple(v.shape) == tuple(shape), (tuple(v.shape), tuple(shape))
        return v

    def rand1(self) -> float:
        return self.uniforms.pop(0)


# ---------------------------------------------------------------------------------------------------
# Random / Offset (noise.py:58-113)
# ---------------------------------------------------------------------------------------------------
def random_noise(shape, randn):
    return randn(shape)


def offset_shape(shape, dims=(0,)):
    "noise.py:105"
    return tuple(d if n in dims else 1 for n, d in enumerate(shape))


def offset_noise(shape, randn, dims=(0,), strength: float = 0.2, static_offset=None):
    """noise.py:104-113.  Draw order: offset first (unless static), then the full-size normal."""
    off = static_offset if static_offset is not None else randn(offset_shape(shape, dims)) * strength**2
    return randn(shape) + off


# ---------------------------------------------------------------------------------------------------
# Pyramid (noise.py:146-207)
# ---------------------------------------------------------------------------------------------------
def pyramid_levels(shape, rand1: Callable[[], float], dims=(-1, -2)):
    """The level geometry alone (noise.py:148-162,195-196): yields the cumulative running shape of
    every level.  One uniform draw per level; the shrink of level i is r**i applied to the *running*
    shape, so sizes fall off super-geometrically."""
    nd = len(shape)
    on = [(nd + d if d < 0 else d) for d in dims]
    mask = [n in on for n in range(nd)]
    run = list(shape)
    for i in range(99):
        r = rand1() * 2 + 2
        run = [max(1, int(s / (r**i))) if m else s for m, s in zip(mask, run)]
        yield i, tuple(run), mask
        if any(s <= 1 for m, s in zip(mask, run) if m):
            break


def upsample_level(v: torch.Tensor, shape, mask) -> torch.Tensor:
    "noise.py:167-193: permute resized dims last, interpolate slice by slice, permute back"
    target = tuple(s for m, s in zip(mask, shape) if m)
    mode = ["linear", "bilinear", "bicubic"][len(target) - 1]
    order = sorted(zip(mask, range(len(shape)), list(v.shape)), key=lambda t: t[0])
    pmask, pdims, pshape = [t[0] for t in order], [t[1] for t in order], [t[2] for t in order]
    lead = pmask.index(True)
    compact = (math.prod(pshape[:lead]), *pshape[lead:])
    v = v.permute(pdims).reshape(compact)
    v = torch.stack([torch.nn.functional.interpolate(s.unsqueeze(0).unsqueeze(0), target, mode=mode).squeeze().squeeze() for s in v])
    back = torch.tensor(pdims, dtype=torch.int).argsort().tolist()
    v = v.reshape([compact[0], *target] if lead > 0 else target).permute(back)
    return v.reshape(shape)


def pyramid_component(shape, randn, rand1, dims=(-1, -2), strength: float = 0.3, depth: int = 99, dtype=torch.float32):
    "noise.py:146-200: sum over levels of strength**i * upsample(randn(level shape))"
    acc = torch.zeros(tuple(shape), dtype=dtype)
    levels = []
    for i, run, mask in pyramid_levels(shape, rand1, dims):
        levels.append(upsample_level(randn(run), tuple(shape), mask) * strength**i)
    n = len(levels) - 1
    skip = min(n, max(0, n - depth))
    return acc + sum(levels[skip:])


def pyramid_noise(shape, randn, rand1, dims=(-1, -2), strength: float = 0.3, depth: int = 99, static_pyramid=None):
    "noise.py:202-207: base normal is drawn FIRST, then the pyramid; unbiased std over the whole unit"
    base = randn(shape)
    pyr = static_pyramid if static_pyramid is not None else pyramid_component(shape, randn, rand1, dims, strength, depth, base.dtype)
    n = base + pyr
    return n / n.std()


# ---------------------------------------------------------------------------------------------------
# Colored (noise.py:284-425)
# ---------------------------------------------------------------------------------------------------
def radial_freq_grid(shape) -> torch.Tensor:
    "noise.py:284-335: normalised radius of every rfftn bin"
    nd = len(shape)
    axes = []
    for i, d in enumerate(shape):
        if i == nd - 1:
            axes.append(torch.arange(d // 2 + 1) / d)
        else:
            axes.append(torch.fft.fftfreq(d, d=1.0).abs())
    rad = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).norm(p=2, dim=-1)
    top = rad.max()
    if top > 0:
        rad = rad / top
    return rad


def colorize(white: torch.Tensor, exponent: float = 0.0, energy: float | None = None) -> torch.Tensor:
    "noise.py:337-405"
    wstd = white.std()
    if exponent == 0.0:
        return white if energy is None or wstd < 1e-8 else white * (energy / wstd)
    w = white.squeeze()
    if w.dtype not in (torch.float32, torch.float64):
        w = w.to(torch.float32)
    spec = torch.fft.rfftn(w)
    grid = radial_freq_grid(w.shape)
    n_eff = sum(w.shape) / len(w.shape) if w.shape else 1.0
    clip = 0.5 / max(n_eff, 4.0)
    weights = torch.clamp(grid, min=clip) ** (-exponent / 2.0)
    col = torch.fft.irfftn(spec * weights, s=w.shape)
    cstd = col.std()
    if cstd > 1e-8:
        col *= wstd / cstd if energy is None else energy / cstd
    return col.view(white.shape).to(dtype=white.dtype)


def colored_exponent(step, color_start: float = 1 / 4, color_end: float = -2, color_curve: float = 2) -> float:
    "noise.py:410-420"
    if step is None:
        return color_start
    if color_curve == math.inf:
        return color_end
    t = stp_clamp(stp_normal(step))[1]
    shift = rescale_positive(-color_curve)
    t = shift / (shift + (divf(1, t) - 1))
    return (1 - t) * color_start + t * color_end


def colored_noise(shape, randn, step, energy=None, color_start: float = 1 / 4, color_end: float = -2, color_curve: float = 2):
    "noise.py:407-425"
    return colorize(randn(shape), colored_exponent(step, color_start, color_end, color_curve), energy)


def batch(gen_one: Callable[[int], torch.Tensor], n: int) -> torch.Tensor:
    "noise.py:445-446: independent per-sample generators, stacked"
    return torch.stack([gen_one(i) for i in range(n)])


# ---------------------------------------------------------------------------------------------------
# Philox4x32-10 + Box-Muller: specification of the device RNG
# ---------------------------------------------------------------------------------------------------
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(ctr: np.ndarray, key: np.ndarray, rounds: int = 10) -> np.ndarray:
    """ctr[...,4] uint32, key[...,2] uint32 (broadcastable) -> [...,4] uint32.
    Round: (c0,c1,c2,c3) <- (hi(M1*c2)^c1^k0, lo(M1*c2), hi(M0*c0)^c3^k1, lo(M0*c0)); key += W."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = c[0].astype(np.uint64) * PHILOX_M0
            p1 = c[2].astype(np.uint64) * PHILOX_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK32).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + PHILOX_W0).astype(np.uint32)
            k1 = (k1 + PHILOX_W1).astype(np.uint32)
    return np.stack(np.broadcast_arrays(*c), axis=-1)


TWO_POW_M32 = np.float32(2.0**-32)
TWO_PI = np.float32(6.283185307179586)


def box_muller(u32: np.ndarray) -> np.ndarray:
    """[...,4] uint32 -> [...,4] float32 standard normals.
    u = (x + 0.5) * 2^-32 evaluated in float32 via fma-free arithmetic: u = float(x)*2^-32 + 2^-33
    (so u in (0,1]); pairs (u0,u1),(u2,u3): r = sqrt(-2 ln u0); (r cos 2pi u1, r sin 2pi u1)."""
    x = u32.astype(np.float32) * TWO_POW_M32 + np.float32(2.0**-33)
    x = np.minimum(x, np.float32(1.0))
    out = np.empty(x.shape, dtype=np.float32)
    for a, b in ((0, 1), (2, 3)):
        r = np.sqrt(np.float32(-2.0) * np.log(x[..., a])).astype(np.float32)
        th = (TWO_PI * x[..., b]).astype(np.float32)
        out[..., a] = r * np.cos(th)
        out[..., b] = r * np.sin(th)
    return out


def philox_normal(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float32 normals for one sample: element e comes from counter block e//4, lane e%4.
    key = (seed lo32, seed hi32); ctr = (block lo32, block hi32, stream lo32, stream hi32).
    `stream` numbers independent draws of the same sample (step index, pyramid level, ...)."""
    blocks = np.arange(offset // 4, (offset + n + 3) // 4 + 1, dtype=np.uint64)
    ctr = np.stack(
        [
            (blocks & _MASK32).astype(np.uint32),
            (blocks >> np.uint64(32)).astype(np.uint32),
            np.full(blocks.shape, stream & 0xFFFFFFFF, dtype=np.uint32),
            np.full(blocks.shape, (stream >> 32) & 0xFFFFFFFF, dtype=np.uint32),
        ],
        axis=-1,
    )
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    flat = box_muller(philox4x32(ctr, key)).reshape(-1)
    start = offset - (offset // 4) * 4
    return flat[start : start + n]


# ---- Brownian (reference noise.py:210-242; tree from torchsde, see the module header) -------------------
BROWNIAN_STREAMS = 1 << 63


def brownian_depth(max_steps: int = 10_000) -> int:
    return math.ceil(math.log2(max_steps * 10)) + 2


def brownian_value(seed: int, n: int, t: float, depth: int) -> np.ndarray:
    """W(t) for n elements of one sample (float64), by explicit bisection: W(0) = 0, W(1) = Z_0 and for the dyadic
    interval with heap index h (root 1, children 2h, 2h+1): W(mid) = (W(a)+W(b))/2 + sqrt(b-a)/2 * Z_h, with
    Z_h = philox_normal(seed, 2^63 | h); linear inside a leaf after `depth` levels."""
    lo, hi, node = 0.0, 1.0, 1
    w_lo = np.zeros(n)
    w_hi = philox_normal(seed, BROWNIAN_STREAMS | 0, n).astype(np.float64)
    for _ in range(depth):
        if t == lo or t == hi:
            break
        mid = 0.5 * (lo + hi)
        w_mid = 0.5 * (w_lo + w_hi) + 0.5 * math.sqrt(hi - lo) * philox_normal(seed, BROWNIAN_STREAMS | node, n).astype(np.float64)
        if t < mid:
            hi, w_hi, node = mid, w_mid, 2 * node
        else:
            lo, w_lo, node = mid, w_mid, 2 * node + 1
    f = (t - lo) / (hi - lo)
    return (1 - f) * w_lo + f * w_hi


def brownian_grid_value(seed: int, n: int, t: float, grid: int, depth: int) -> np.ndarray:
    """W(t) on the path built over the `grid`-cell partition of [0, 1] (the product's round-4 construction, restated on arrays):
    grid points by bisection of the index range -- W(1) = Z_0 and, for the index interval (lo, hi) with heap index h and
    mid = (lo + hi) // 2,  W(mid/N) = ((hi-mid) W(lo/N) + (mid-lo) W(hi/N)) / (hi-lo) + sqrt((mid-lo)(hi-mid) / ((hi-lo) N)) Z_h,
    Z_h = philox_normal(seed, 2^63 | 2^61 | h) -- then a dyadic bridge inside the cell, Z = philox_normal(seed, 2^63 | 2^60 | cell << 24 | h)."""

    def point(j: int) -> np.ndarray:
        lo, hi, node = 0, grid, 1
        w_lo, w_hi = np.zeros(n), philox_normal(seed, BROWNIAN_STREAMS | 0, n).astype(np.float64)
        while True:
            if j == lo:
                return w_lo
            if j == hi:
                return w_hi
            mid = (lo + hi) // 2
            z = philox_normal(seed, BROWNIAN_STREAMS | (1 << 61) | node, n).astype(np.float64)
            w_mid = ((hi - mid) * w_lo + (mid - lo) * w_hi) / (hi - lo) + math.sqrt((mid - lo) * (hi - mid) / ((hi - lo) * grid)) * z
            if j < mid:
                hi, w_hi, node = mid, w_mid, 2 * node
            else:
                lo, w_lo, node = mid, w_mid, 2 * node + 1

    j = round(t * grid)
    if 0 <= j <= grid and j / grid == t:
        return point(j)
    cell = min(int(t * grid), grid - 1)
    if not cell / grid < t < (cell + 1) / grid:
        cell = cell - 1 if t < cell / grid else cell + 1
    lo, hi, node = cell / grid, (cell + 1) / grid, 1
    w_lo, w_hi = point(cell), point(cell + 1)
    for _ in range(max(depth - (grid - 1).bit_length(), 2)):
        if t == lo or t == hi:
            break
        mid = 0.5 * (lo + hi)
        z = philox_normal(seed, BROWNIAN_STREAMS | (1 << 60) | (cell << 24) | node, n).astype(np.float64)
        w_mid = 0.5 * (w_lo + w_hi) + 0.5 * math.sqrt(hi - lo) * z
        if t < mid:
            hi, w_hi, node = mid, w_mid, 2 * node
        else:
            lo, w_lo, node = mid, w_mid, 2 * node + 1
    f = (t - lo) / (hi - lo)
    return (1 - f) * w_lo + f * w_hi


def brownian_noise(seed: int, shape, step, max_steps: int = 10_000, grid: int | None = None) -> torch.Tensor:
    """Brownian.generate (noise.py:238-242): step.normal().clamp(), then tree(t0, t1) / sqrt(distance); fp64 result.
    `grid` = N: the generator's path is the one built over the N-cell partition (what a generator first asked Step.from_int(k, N) uses)."""
    t0, t1 = stp_clamp(stp_normal(step))
    n, depth = math.prod(shape), brownian_depth(max_steps)
    value = (lambda t: brownian_grid_value(seed, n, t, grid, depth)) if grid else (lambda t: brownian_value(seed, n, t, depth))
    inc = value(t1) - value(t0)
    return torch.from_numpy(inc / math.sqrt(t1 - t0)).reshape(tuple(shape))
