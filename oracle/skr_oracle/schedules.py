"""Sigma schedules as closures over numpy fp64.  Follows reference skrample/scheduling.py.

A schedule here is a `Sched`: a function t[N] -> rows (timestep, sigma, alpha) with t=1 all noise,
plus the sigma space tag ("vp" = variance preserving, "fm" = flow matching) and, for base
schedules, the inverse map sigma -> row used by sub-sigma schedules (Karras & co).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable

import numpy as np

from .scalars import Pt, rescale_positive

Arr = np.ndarray


# ---- sigma spaces (scheduling.py:31-48) ---------------------------------------------------------
def space_normalize(space: str, regular):
    "regular sigma -> (sigma, alpha)"
    if space == "vp":  # scheduling.py:33-35
        th = np.arctan(regular)
        return np.sin(th), np.cos(th)
    reg = np.asarray(regular)  # scheduling.py:43-45
    return reg, 1 - reg


def space_regularize(space: str, normal):
    "normalised sigma -> regular sigma"
    if space == "vp":  # scheduling.py:37-38
        return np.tan(np.arcsin(normal))
    return np.asarray(normal)  # scheduling.py:47-48


@dataclass(frozen=True)
class Sched:
    fn: Callable[[Arr], Arr] = field(compare=False)
    space: str
    base_timesteps: int = 1000
    sig2pts: Callable[[Arr, Arr], Arr] | None = field(default=None, compare=False)
    tag: str = ""

    # scheduling.py:79-107
    def points_np(self, t) -> Arr:
        return self.fn(np.asarray(t, dtype=np.float64).clip(0, 1))

    def ipoints_np(self, t) -> Arr:
        return self.fn(1 - np.asarray(t, dtype=np.float64).clip(0, 1))

    def points(self, t) -> list[Pt]:
        return [Pt(*r) for r in self.points_np(t).tolist()]

    def ipoints(self, t) -> list[Pt]:
        return [Pt(*r) for r in self.ipoints_np(t).tolist()]

    def point(self, t: float) -> Pt:
        return Pt(*self.fn(np.expand_dims(np.float64(t).clip(0, 1), 0))[0].tolist())

    def ipoint(self, t: float) -> Pt:
        return Pt(*self.fn(np.expand_dims(1 - np.float64(t).clip(0, 1), 0))[0].tolist())

    # scheduling.py:129-135
    def schedule_np(self, steps: int) -> Arr:
        return self.fn(np.linspace(1, 0, steps, endpoint=False))

    def schedule(self, steps: int) -> tuple[Pt, ...]:
        return tuple(Pt(*r) for r in self.schedule_np(steps).tolist())


def _tcol(t: Arr, base_timesteps: int) -> Arr:
    "timestep column; negative base flips direction (scheduling.py:242,302)"
    return ((1 - t) if base_timesteps < 0 else t) * abs(base_timesteps)


# ---- Scaled / ZSNR (scheduling.py:180-278) -------------------------------------------------------
def _scaled_acp(t: Arr, beta_start: float, beta_end: float, k: float, T: int) -> Arr:
    "closed-form continuous alphas_cumprod, scheduling.py:192-229"
    r0 = beta_start ** (1 / k)
    r1 = beta_end ** (1 / k)
    m = r1 - r0
    if abs(m) < 1e-8:
        b = r0**k
        i1 = b * t
        i2 = (b**2) * t
    else:
        i1 = ((r0 + m * t) ** (k + 1) - r0 ** (k + 1)) / (m * (k + 1))
        i2 = ((r0 + m * t) ** (2 * k + 1) - r0 ** (2 * k + 1)) / (m * (2 * k + 1))
    return np.exp(-(T * (i1 + i2 / 2)))


def _zsnr_acp(t: Arr, beta_start: float, beta_end: float, k: float, T: int) -> Arr:
    "zero-terminal-SNR rescale, scheduling.py:258-278"
    root = np.sqrt(_scaled_acp(np.concatenate([[0], t, [1]]), beta_start, beta_end, k, T))
    first, last = root[0].item(), root[-1].item()
    root = root[1:-1]
    root -= last
    root *= first / (first - last)
    return root**2


def _vp_base(acp_fn, tag: str, base_timesteps: int, beta_start: float, beta_end: float, beta_scale: float) -> Sched:
    T = abs(base_timesteps)

    def fn(t: Arr) -> Arr:  # scheduling.py:231-246
        acp = acp_fn(t, beta_start, beta_end, beta_scale, T)
        with np.errstate(divide="ignore"):
            sig = np.sqrt((1 - acp) / acp)
        return np.stack([_tcol(t, base_timesteps), *space_normalize("vp", sig)], 1)

    cache: dict[str, Arr] = {}

    def all_points() -> Arr:  # scheduling.py:147-153
        if "p" not in cache:
            n = T if T > 1 else 10_000
            cache["p"] = fn(np.linspace(0, 1, n).clip(0, 1))
        return cache["p"]

    def sig2pts(sig: Arr, alp: Arr) -> Arr:  # scheduling.py:248-251
        ap = all_points()
        return np.stack([np.interp(sig, ap[:, 1], ap[:, 0]), sig, alp], axis=1)

    return Sched(fn, "vp", base_timesteps, sig2pts, f"{tag}({base_timesteps},{beta_start},{beta_end},{beta_scale})")


def scaled(base_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012, beta_scale: float = 2) -> Sched:
    return _vp_base(_scaled_acp, "scaled", base_timesteps, beta_start, beta_end, beta_scale)


def zsnr(base_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012, beta_scale: float = 2) -> Sched:
    return _vp_base(_zsnr_acp, "zsnr", base_timesteps, beta_start, beta_end, beta_scale)


# ---- Linear (scheduling.py:281-317) ---------------------------------------------------------------
def linear(base_timesteps: int = 1000, sigma_start: float = 1, custom_space: str | None = None) -> Sched:
    space = custom_space or ("fm" if sigma_start <= 1 else "vp")

    def fn(t: Arr) -> Arr:
        return np.stack([_tcol(t, base_timesteps), *space_normalize(space, t * sigma_start)], axis=1)

    def sig2pts(sig: Arr, alp: Arr) -> Arr:
        tt = ((sigma_start - sig) if base_timesteps < 0 else sig) * (abs(base_timesteps) / sigma_start)
        return np.stack([tt, sig, alp], axis=1)

    return Sched(fn, space, base_timesteps, sig2pts, f"linear({base_timesteps},{sigma_start},{space})")


# ---- sub-sigma schedules (scheduling.py:364-380, 493-580) -----------------------------------------
def _sub(base: Sched, sub_sigmas: Callable[[Arr], Arr], tag: str) -> Sched:
    assert base.sig2pts is not None, "sub-sigma schedules need a base schedule"

    def fn(t: Arr) -> Arr:
        return base.sig2pts(*space_normalize(base.space, sub_sigmas(t)))

    return Sched(fn, base.space, base.base_timesteps, None, f"{tag}<{base.tag}>")


def _reg(base: Sched, t: float) -> float:
    return space_regularize(base.space, base.point(t).s).item()


def _renorm(x: Arr, hi, lo=0):
    "common.py:153-155 `normalize`"
    return (x - lo) / (hi - lo)


def karras(base: Sched, rho: float = 7.0, steps: float = 20) -> Sched:
    def sub(t: Arr) -> Arr:  # scheduling.py:507-514
        smin, smax = _reg(base, 1 / steps), _reg(base, 1)
        tt = np.concatenate([[1, 0], t])
        s = ((smin ** (1.0 / rho)) * (1 - tt) + (smax ** (1.0 / rho)) * tt) ** rho
        return _renorm(s[2:], s[0], s[1]) * smax

    return _sub(base, sub, f"karras({rho},{steps})")


def exponential(base: Sched, rho: float = 1.0, steps: float = 20) -> Sched:
    def sub(t: Arr) -> Arr:  # scheduling.py:531-538
        smin, smax = _reg(base, 1 / steps), _reg(base, 1)
        tt = np.concatenate([[1, 0], t]) ** rho
        s = np.exp(np.log(smin) * (1 - tt) + np.log(smax) * tt)
        return _renorm(s[2:], s[0], s[1]) * smax

    return _sub(base, sub, f"exponential({rho},{steps})")


def beta(base: Sched, alpha: float = 0.6, beta_: float = 0.6) -> Sched:
    def sub(t: Arr) -> Arr:  # scheduling.py:549-558
        from scipy.stats import beta as beta_dist

        smax = _reg(base, 1)
        s = beta_dist.ppf(np.concatenate([[1], t]), alpha, beta_)
        return _renorm(s, s[0])[1:] * smax

    return _sub(base, sub, f"beta({alpha},{beta_})")


def probit(base: Sched, scale: float = 3) -> Sched:
    def sub(t: Arr) -> Arr:  # scheduling.py:570-580
        from scipy.stats import norm

        tt = np.concatenate([[1, 0], t])
        prob = tt * ((1 - 1e-8) - 0) + 0  # common.py:158-160 `regularize`
        e = math.e ** norm.ppf(prob, scale=scale)  # common.py:173-179 sigmoid via math.e ** x
        s = e / (1 + e)
        return _renorm(s[2:], *s[:2]) * _reg(base, 1)

    return _sub(base, sub, f"probit({scale})")


# ---- time modifiers (scheduling.py:383-395, 583-664) ----------------------------------------------
def _mod(base: Sched, modify: Callable[[Arr], Arr], tag: str) -> Sched:
    return Sched(lambda t: base.fn(modify(t)), base.space, base.base_timesteps, None, f"{tag}<{base.tag}>")


def flowshift(base: Sched, shift: float = 3.0) -> Sched:
    return _mod(base, lambda t: shift * t / (1 + (shift - 1) * t), f"flowshift({shift})")  # :588-592


def hyper(base: Sched, scale: float = 2, tail: bool = True) -> Sched:
    def modify(t: Arr) -> Arr:  # scheduling.py:606-614
        if abs(scale) <= 1e-8:
            return t
        lo = -scale * tail
        p = np.concatenate([[1], t]) * (scale - lo) + lo
        p = np.sinh(p) if scale < 0 else np.tanh(p / math.sqrt(2))
        return _renorm(p[1:], p[0], -p[0] * tail)

    return _mod(base, modify, f"hyper({scale},{tail})")


def sinner(base: Sched, count: float = -2, scale: float = 2) -> Sched:
    def modify(t: Arr) -> Arr:  # scheduling.py:635-664
        if abs(scale) <= 1e-8 or count == math.inf:
            return t
        n = rescale_positive(count * 2 ** math.copysign(1, count)) + 1
        tt = np.concatenate([[0, 1], 1 - t])
        period = tt * (math.pi * n)
        if scale >= 0:
            period += math.pi
        k = abs(scale) ** -1 + 1
        p = np.sin(period) + period * k
        return _renorm(p[2:], *p[:2])

    return _mod(base, modify, f"sinner({count},{scale})")


def fixed(rows, space: str) -> Sched:
    "scheduling.py:160-177 FixedSchedule: linear interpolation through given rows + trailing (0,0,1)"
    from scipy.interpolate import make_interp_spline

    tab = np.concatenate([np.asarray(rows, dtype=np.float64), [[0, 0, 1]]])
    spline = make_interp_spline(np.linspace(0, 1, len(tab)), tab, k=1, axis=0)
    return Sched(lambda t: spline(1 - t), space, 1000, None, "fixed")


def fixed_from_regular(timesteps, regular_sigmas, space: str) -> Sched:
    "scheduling.py:165-167"
    return fixed(np.stack([timesteps, *space_normalize(space, regular_sigmas)], axis=1), space)
