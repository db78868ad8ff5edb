"""Explicit Runge-Kutta tableau stepping.  Follows reference skrample/sampling/functional.py:55-105
(step_tableau), :217-349 (RKUltra, DynasauRK), skrample/sampling/tableaux/* (the few tableaux the
default provider map needs) and skrample/diffusers.py:746-873 (the inside-out stage machine).

A tableau is (nodes, weights): nodes = ((c_j, (a_j0..a_j,j-1)), ...), weights = (b_0..b_s-1).
"""

from __future__ import annotations

import math
from fractions import Fraction as Fr

from . import predictors as P
from .scalars import Pt, Stp, stp_amount, stp_clamp, stp_from_int, stp_normal, stp_position, sumprod
from .schedules import Sched

R2 = math.sqrt(2)


# ---- tableaux ------------------------------------------------------------------------------------
def tab_rk2(c1: float):
    "providers.py:15-23"
    return ((0.0, ()), (c1, (c1,))), (1 - 1 / (2 * c1), 1 / (2 * c1))


def tab_ees25(x: float):
    "providers.py:83-94  EES(2,5;x), arXiv 2507.21006"
    c1 = (1 + 2 * x) / (4 * (1 - x))
    return (
        (0.0, ()),
        (c1, (c1,)),
        (3 / (4 * (1 - x)), ((4 * x - 1) ** 2 / (4 * (x - 1) * (1 - 4 * x**2)), (1 - x) / (1 - 4 * x**2))),
    ), (x, 1 / 2, 1 / 2 - x)


def tab_ees27(x: float):
    "providers.py:97-127  EES(2,7;x)"
    A = (2 * x + R2) / ((2 * x - 1) * (-2 * x - R2 + 1))
    B = 1 / ((2 * x - 1) * (1 - R2 - 2 * x) * (2 - R2 - 2 * x))
    a2 = ((-2 + R2 * (1 - 2 * x)) / (4 * (x - 1)),)
    a3 = ((((2 * x + R2 - 2) * (4 * x + R2 - 2)) / (4 * R2 * (x - 1))) * A, (0.5 * (-1 + R2)) * A)
    a4 = (
        ((2 * x - R2) * (-40 * x**4 + (80 - 40 * R2) * x**3 - (88 - 60 * R2) * x**2 + (48 - 34 * R2) * x + 7 * R2 - 10))
        / (4 * (x - 1) * (2 * x**2 - 1))
        * B,
        (2 - R2) * x * (x - 1) * (4 * x + R2 - 2) * B,
        ((2 - R2) * (2 * x - R2) * (2 + R2 - 2 * x) * (x - 1) * (2 * x - 1)) / (4 * (2 * x**2 - 1) * (2 * x**2 - 4 * x + 1)),
    )
    return ((0.0, ()), (math.fsum(a2), a2), (math.fsum(a3), a3), (math.fsum(a4), a4)), (
        x,
        1 / 2 * (2 - R2) - (1 - R2) * x,
        (1 - R2) * (x - 1),
        1 / 2 * (2 - R2) - x,
    )


def tab_shu_osher(alphas, betas):
    "tableaux/common.py:99-121: Shu-Osher (alpha, beta) form -> Butcher form"
    s = len(alphas)
    a = [[-math.inf] * n for n in range(s)]
    for i in range(1, s):
        for j in range(i):
            a[i][j] = math.fsum((betas[i - 1][j], *(alphas[i - 1][k] * a[k][j] for k in range(j + 1, i))))
    b = [math.fsum((betas[s - 1][j], *(alphas[s - 1][k] * a[k][j] for k in range(j + 1, s)))) for j in range(s)]
    return tuple((math.fsum(row), tuple(row)) for row in a), tuple(b)


def _fr(rows, weights):
    return tuple((float(Fr(c)), tuple(float(Fr(v)) for v in a)) for c, a in rows), tuple(float(Fr(w)) for w in weights)


TAB_EULER = ((0, ()),), (1,)  # providers.py:175-178
TAB_HEUN = ((0, ()), (1, (1,))), (1 / 2, 1 / 2)  # providers.py:391-398 (embedded row dropped)
TAB_SSPRK4_5 = tab_shu_osher(  # providers.py:578-593, Ruuth 2006
    [
        [1],
        [0.444370493651235, 0.555629506348765],
        [0.620101851488403, 0, 0.379898148511597],
        [0.178079954393132, 0, 0, 0.821920045606868],
        [0, 0, 0.517231671970585, 0.096059710526147, 0.386708617503269],
    ],
    [
        [0.391752226571890],
        [0, 0.368410593050371],
        [0, 0, 0.251891774271694],
        [0, 0, 0, 0.544974750228521],
        [0, 0, 0, 0.063692468666290, 0.226007483236906],
    ],
)
# providers.py:461-472 Cash-Karp; python evaluates e.g. 1631/55296 as one float division, so plain
# float division of the integer ratios reproduces the reference constants bit for bit.
TAB_CASHKARP = (
    (
        (0, ()),
        (1 / 5, (1 / 5,)),
        (3 / 10, (3 / 40, 9 / 40)),
        (3 / 5, (3 / 10, -9 / 10, 6 / 5)),
        (1, (-11 / 54, 5 / 2, -70 / 27, 35 / 27)),
        (7 / 8, (1631 / 55296, 175 / 512, 575 / 13824, 44275 / 110592, 253 / 4096)),
    ),
    (37 / 378, 0, 250 / 621, 125 / 594, 0, 512 / 1771),
)

DEFAULT_BY_STAGES = {  # functional.py:18-30 (entries 1..6 only)
    1: TAB_EULER,
    2: tab_rk2(1 / 2),
    3: tab_ees25(1 / 10),
    4: tab_ees27(1 / 14 * (5 - 3 * R2)),
    5: TAB_SSPRK4_5,
    6: TAB_CASHKARP,
}


def pick_tableau(order: int, providers=None):
    "functional.py:231-238: largest provider key <= order, else Euler"
    providers = DEFAULT_BY_STAGES if providers is None else providers
    if order >= min(providers) and (m := max(o for o in providers if o <= order)):
        return providers[m]
    return TAB_EULER


def adjust_steps(tab, steps: int) -> int:
    "functional.py:240-247"
    nodes = tab[0]
    return max(round(steps / len(nodes) + sum(abs(1 - c) < 1e-8 for c, _ in nodes) / len(nodes)), 1)


# ---- the step -------------------------------------------------------------------------------------
def step_tableau(tab, x, model, pred, sched: Sched, step, deriv=None, noise=None, eta: float = 0, epsilon: float = 1e-8):
    "functional.py:55-105 (single weight row)"
    nodes, weights = tab
    if deriv:
        raw = model

        def model(xx, t, s, a, _raw=raw, _pred=pred):  # models.py:232-239 wrap_model_call
            return P.convert(_pred, deriv, xx, _raw(xx, t, s, a), Pt(t, s, a))

        pred = deriv

    ders = []
    s0, s1, *fracs = sched.ipoints([*step, *(step[0] + c * (step[1] - step[0]) for c, _ in nodes)])
    for sn, (_, arow) in zip(fracs, nodes):
        if arow:
            xi = P.forward(pred, x, sumprod(ders, arow) / math.fsum(arow), s0, sn)
        else:
            xi = x
        if abs(sn.t) < epsilon or abs(sn.s) < epsilon:
            ders.append(P.backward(pred, x, xi, s0, s1))
        else:
            ders.append(model(xi, *sn))
    return P.forward(pred, x, sumprod(ders, weights), s0, s1, noise, eta)


def rk_loop(tab_for_step, x, model, pred, sched: Sched, steps: int, include=slice(None), rng=None, callback=None, deriv="data", eta: float = 0):
    "functional.py:176-194 + 249-268 / 330-349"
    for n in list(range(steps))[include]:
        step = stp_from_int(n, steps)
        x = step_tableau(tab_for_step(step), x, model, pred, sched, step, deriv, rng(step) if rng else None, eta)
        if callback:
            callback(x, n, sched.ipoints(step))
    return x


def dynasaur_tableau(step, order: int = 2, per_step_decay=math.log(0.5) / -2, total_step_decay=math.log(0.5) / -20, invert=False):
    "functional.py:302-328"
    if order >= 4:
        high, low, tf = 1 / 4 * (2 - R2), 1 / 14 * (5 - 3 * R2), tab_ees27
    elif order >= 3:
        high, low, tf = 0.25, 0.1, tab_ees25
    else:
        high, low, tf = 1, 0.5, tab_rk2
    stages = len(tf((high + low) / 2)[0])
    st = stp_clamp(stp_normal(step))
    g = math.exp((-total_step_decay * stp_amount(st) - per_step_decay * stp_position(st)) * stages)
    g = abs(invert - min(max(g, 0), 1))
    return tf(g * high + (1 - g) * low)


# ---- inside-out stage machine (diffusers.py:602-873) -------------------------------------------------
def rk_all_points(tab, sched: Sched, steps: int, epsilon: float = -math.inf) -> list[Pt]:
    """diffusers.py:943-963: every point the model is evaluated at (t=0 stages included).  `tab` may be a function of the step (DynasauRK,
    diffusers.py:1029-1042: there the table comes out of a scalar run of the functional sampler itself, default epsilon -- stages on the clean
    end are not recorded and the count assertion fires for schedules that put one there)."""
    seen: list[Pt] = []

    def rec(x, t, s, a):
        seen.append(Pt(t, s, a))
        return x

    for n in range(steps):
        step = stp_from_int(n, steps)
        step_tableau(tab(step) if callable(tab) else tab, 1, rec, "data", sched, step, epsilon=epsilon)
    return seen


class InsideOutRK:
    """State machine equivalent to RKWrapperCore.step / step_tableau_inside_out for a fixed tableau.
    One `feed(sample, output)` per model evaluation; returns the next model input (or the step result)."""

    def __init__(self, tab, sched: Sched, steps: int, pred, deriv="data", eta: float = 0):
        self.tab, self.sched, self.steps, self.pred, self.deriv, self.eta = tab, sched, steps, pred, deriv, eta
        self.order = len((tab(stp_from_int(0, 1)) if callable(tab) else tab)[0])
        if callable(tab):  # DynasauRK: a tableau per step (diffusers.py:1023-1042)
            self.all_points = rk_all_points(tab, sched, steps, epsilon=1e-8)
            assert len(self.all_points) == self.order * steps
        else:
            self.all_points = rk_all_points(tab, sched, steps)
        self.index = 0
        self.ders: list = []
        self.base = None
        self.p0 = sched.point(0)

    def _stage(self, sample, output, space, s0, s1, sn, noise_fn):
        "diffusers.py:746-796"
        nodes, weights = self.tab(stp_from_int(self.index // self.order, self.steps)) if callable(self.tab) else self.tab
        self.ders.append(output)
        if self.base is None:
            self.base = sample
        base = self.base
        if len(self.ders) == len(weights):
            noise = noise_fn(stp_from_int(self.index // self.order, self.steps)) if abs(self.eta) > 1e-8 else None
            out = P.forward(space, base, sumprod(self.ders, weights), s0, s1, noise, self.eta)
            self.ders.clear()
            self.base = None
            return out
        arow = nodes[len(self.ders)][1]
        if arow:
            return P.forward(space, base, sumprod(self.ders, arow) / math.fsum(arow), s0, sn)
        raise ValueError

    def feed(self, sample, output, noise_fn=None, cast=lambda v: v):
        "diffusers.py:798-873 (timestep assertion left to the caller)"
        pts = [*self.all_points, Pt(0, 0, 1)]
        if self.deriv:
            output = P.convert(self.pred, self.deriv, sample, output, pts[self.index])
            space = self.deriv
        else:
            space = self.pred
        i0 = self.index - len(self.ders)
        i1 = self.index + self.order - len(self.ders)
        res = self._stage(cast(sample), cast(output), space, pts[i0], pts[i1], pts[self.index + 1], noise_fn)
        self.index += 1
        while self.index < len(self.all_points) and (
            abs(self.all_points[self.index].t - self.p0.t) < 1e-8 or abs(self.all_points[self.index].s - self.p0.s) < 1e-8
        ):
            synth = P.backward(space, cast(sample if self.base is None else self.base), res, pts[i0], pts[i1])
            res = self._stage(cast(sample), synth, space, pts[i0], pts[i1], pts[self.index + 1], noise_fn)
            self.index += 1
        return res


# ---- adaptive Runge-Kutta (functional.py:197-214, 352-472) ----------------------------------------------------
# embedded pairs: (nodes, weights, error_weights)   providers.py:391-398, 418-427, 449-460
EMB_HEUN = (((0, ()), (1, (1,))), (1 / 2, 1 / 2), (1, 0))
EMB_BOGACKI_SHAMPINE = (
    ((0, ()), (1 / 2, (1 / 2,)), (3 / 4, (0, 3 / 4)), (1, (2 / 9, 1 / 3, 4 / 9))),
    (2 / 9, 1 / 3, 4 / 9, 0),
    (7 / 24, 1 / 4, 1 / 3, 1 / 8),
)
EMB_FEHLBERG = (
    (
        (0, ()),
        (1 / 4, (1 / 4,)),
        (3 / 8, (3 / 32, 9 / 32)),
        (12 / 13, (1932 / 2197, -7200 / 2197, 7296 / 2197)),
        (1, (439 / 216, -8, 3680 / 513, -845 / 4104)),
        (1 / 2, (-8 / 27, 2, -3544 / 2565, 1859 / 4104, -11 / 40)),
    ),
    (16 / 135, 0, 6656 / 12825, 28561 / 56430, -9 / 50, 2 / 55),
    (25 / 216, 0, 1408 / 2565, 2197 / 4104, -1 / 5, 0),
)
DEFAULT_EMBEDDED = {2: EMB_HEUN, 4: EMB_BOGACKI_SHAMPINE, 6: EMB_FEHLBERG}  # functional.py:46-50


def _mean(v) -> float:
    "common.py:193-198"
    return v if isinstance(v, (float, int)) else v.mean().item()


def mse(a, b) -> float:
    "functional.py:206-209"
    return _mean(abs(a - b) ** 2)


def mae(a, b) -> float:
    "functional.py:201-204"
    return _mean(abs(a - b))


def step_tableau_embedded(emb, x, model, pred, sched: Sched, step, deriv=None):
    "functional.py:55-105 with both weight rows: returns (high, low)"
    nodes, w_hi, w_lo = emb
    hi = step_tableau((nodes, w_hi), x, model, pred, sched, step, deriv)
    lo = step_tableau((nodes, w_lo), x, model, pred, sched, step, deriv)
    return hi, lo


def rkmoire_loop(x, model, pred, sched: Sched, steps: int, include=slice(None), callback=None, order: int = 2, deriv="data", evaluator=mse,
                 threshold: float = 1e-4, initial: float = 1 / 50, maximum: float = 1 / 4, adaption: float = 0.3, discard: float = float("inf"),
                 rescale_init: bool = True, rescale_max: bool = False, providers=None):
    "functional.py:396-472: error-controlled step size over an embedded pair"
    providers = DEFAULT_EMBEDDED if providers is None else providers
    if order >= min(providers) and (m := max(o for o in providers if o <= order)):
        emb = providers[m]
    else:
        emb = EMB_HEUN
    if rescale_init:
        initial *= len(emb[0]) / 2
    if rescale_max:
        maximum *= len(emb[0]) / 2
    step_size = max(round(steps * initial), 1)
    eps = 1e-16
    indices = list(range(steps))[include]
    step = indices[0]
    while step <= indices[-1]:
        step_next = min(step + step_size, indices[-1] + 1)
        if step_next < steps:
            hi, lo = step_tableau_embedded(emb, x, model, pred, sched, (step / steps, step_next / steps), deriv)
            s0, s1, s2 = sched.ipoints_np([step / steps, step_next / steps, (step_next + step_size) / steps])[:, 1].tolist()
            slope = abs(s0 - s1) / abs(s1 - s2)
            error = evaluator(lo, hi) / max(evaluator(0, hi), eps)
            adjustment = (threshold / max(error, eps)) ** adaption / slope
            step_size = max(round(min(step_size * adjustment, steps * maximum)), 1)
            if step_next - step > step_size and 1 / max(adjustment, eps) > discard:
                continue
        else:
            hi = step_tableau((emb[0], emb[1]), x, model, pred, sched, (step / steps, 1), deriv)
        x = hi
        if callback:
            callback(x, step_next - 1, sched.ipoints(stp_from_int(step, steps)))
        step = step_next
    return x
