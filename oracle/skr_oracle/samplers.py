"""Structured (inside-out, one-step-per-call) samplers.  Follows reference
skrample/sampling/structured.py and skrample/sampling/interface.py.

Records are plain tuples `Rec(sample, prediction, step, noise, final)`; sampler configs are dicts
(`kind`, `order`, `eta`, `deriv`, ...).  T may be float, np.ndarray or torch.Tensor.
"""

from __future__ import annotations

import math
from typing import Any, NamedTuple, Sequence

import numpy as np

from . import predictors as P
from .scalars import Stp, bashforth, divf, ln, softmax2, spowf, stp_amount, stp_from_int, stp_position, sumprod
from .schedules import Sched


class Rec(NamedTuple):
    "structured.py:16-40  SampleInput + SKSamples.final"

    sample: Any
    prediction: Any
    step: Stp
    noise: Any = None
    final: Any = None


MAX_ORDER = {"euler": 1, "dpm": 3, "adams": 9, "unip": 9, "unipc": 9}


def make(kind: str, order: int | None = None, eta: float = 0, deriv="data", **kw) -> dict:
    "sampler config; defaults follow traits.py:24-56 (order=2, stochasticity=0, derivative=DataModel)"
    cfg = {"kind": kind, "order": 2 if order is None else order, "eta": eta, "deriv": deriv}
    if kind == "euler":
        cfg["order"] = 1
    if kind in ("unip", "unipc"):
        cfg.setdefault("fast_solve", False)
    if kind == "unipc":
        cfg.setdefault("predictor", None)
    if kind == "spc":  # structured.py:505-517
        cfg.pop("order"), cfg.pop("eta")
        cfg.update(predictor=make("euler"), corrector=make("adams", 4), bias=0, power=1, adaptive=True, invert=False)
    cfg.update(kw)
    return cfg


def require_noise(cfg: dict) -> bool:
    "structured.py:155-156, 462-463, 520-521"
    if cfg["kind"] == "spc":
        return require_noise(cfg["predictor"]) or require_noise(cfg["corrector"])
    own = abs(cfg["eta"]) > 1e-8
    if cfg["kind"] == "unipc" and cfg["predictor"]:
        return own or require_noise(cfg["predictor"])
    return own


def require_previous(cfg: dict) -> int:
    "structured.py:134-135, 466-467, 524-525"
    k = cfg["kind"]
    if k == "euler":
        return 0
    if k == "spc":
        return max(require_previous(cfg["predictor"]), require_previous(cfg["corrector"]) + 1)
    own = max(min(cfg["order"], MAX_ORDER[k]), 1) - 1
    if k == "unipc":
        return max(own + 1, require_previous(cfg["predictor"]) if cfg["predictor"] else 0)
    return own


def effective_order(cfg: dict, step, n_previous: int) -> int:
    "structured.py:137-149: ramp up from the start, drop toward 1 at the end"
    pos = stp_position(step)
    return max(1, min(MAX_ORDER[cfg["kind"]], round(pos + 1), cfg["order"], n_previous + 1, round(stp_amount(step) - pos)))


def _dpoints(sched: Sched, step):
    "structured.py:33-34"
    return sched.ipoints(step)


def _to_derivative(cfg, cur: Rec, pred, sched: Sched, previous: Sequence[Rec], eo: int, same: bool = False):
    """structured.py:207-220 / 304-317 / 356-371: convert current + the (eo-1) newest history entries
    to derivative space.  Newest first.  History is re-converted every call (not cached)."""
    p0 = _dpoints(sched, cur.step)[0]
    deriv = cfg["deriv"]
    if deriv:
        tail = previous[-eo + 1 :] if eo > 1 else previous[0:]  # python: previous[-0:] == whole list
        hist = [P.convert(pred, deriv, r.sample, r.prediction, _dpoints(sched, r.step)[0], same) for r in tail]
        return [P.convert(pred, deriv, cur.sample, cur.prediction, p0, same), *reversed(hist)], deriv
    tail = previous[-eo + 1 :] if eo > 1 else previous[0:]
    return [cur.prediction, *reversed([r.prediction for r in tail])], pred


def _lam(p) -> float:
    return ln(divf(p.a, p.s))


def euler(cfg, cur: Rec, pred, sched: Sched, previous=()):
    "structured.py:167-180"
    p0, p1 = _dpoints(sched, cur.step)
    return P.forward(pred, cur.sample, cur.prediction, p0, p1, cur.noise, cfg["eta"])


def dpm(cfg, cur: Rec, pred, sched: Sched, previous=()):
    "structured.py:195-283  DPM-Solver++ multistep, orders 1-3, in derivative (x-hat-0) space"
    p0, p1 = _dpoints(sched, cur.step)
    eo = effective_order(cfg, cur.step, len(previous))
    preds, space = _to_derivative(cfg, cur, pred, sched, previous, eo)
    q = preds.pop(0)
    if eo >= 2:
        lam, lam_next = _lam(p0), _lam(p1)
        h = abs(lam_next - lam)
        lam_m1 = _lam(sched.ipoint(previous[-1].step[0]))
        r = (lam - lam_m1) / h
        q_m1 = preds.pop(0)
        d10 = (1.0 / r) * (q - q_m1)
        if eo >= 3:
            lam_m2 = _lam(sched.ipoint(previous[-2].step[0]))
            r2 = (lam_m1 - lam_m2) / h
            q_m2 = preds.pop(0)
            d11 = (1.0 / r2) * (q_m1 - q_m2)
            d1 = d10 + (r / (r + r2)) * (d10 - d11)
            d2 = (1.0 / (r + r2)) * (d10 - d11)
            hh = -h
            e = math.expm1(hh)
            c1 = (e / hh - 1.0) / e if e != 0 else 0
            c2 = ((e - hh) / hh**2 - 0.5) / e if e != 0 else 0
            q = q + c1 * d1 + c2 * d2
        else:
            q = q + 0.5 * d10
    return P.forward(space, cur.sample, q, p0, p1, cur.noise, eta=cfg["eta"])


def adams(cfg, cur: Rec, pred, sched: Sched, previous=()):
    "structured.py:294-330  Adams-Bashforth weighted derivative"
    eo = effective_order(cfg, cur.step, len(previous))
    p0, p1 = _dpoints(sched, cur.step)
    preds, space = _to_derivative(cfg, cur, pred, sched, previous, eo)
    q = sumprod(preds[:eo], bashforth(eo))
    return P.forward(space, cur.sample, q, p0, p1, cur.noise, cfg["eta"])


def unisolve(cfg, cur: Rec, pred, sched: Sched, previous=(), prediction_next=None, same: bool = False):
    "structured.py:344-436  UniP (prediction_next None) / UniC (prediction_next given)"
    p0, p1 = _dpoints(sched, cur.step)
    eo = effective_order(cfg, cur.step, len(previous))
    preds, space = _to_derivative(cfg, cur, pred, sched, previous, eo, same)
    if cfg["deriv"] and prediction_next is not None:
        prediction_next = P.convert(pred, cfg["deriv"], cur.sample, prediction_next, p0, same)
    q = preds.pop(0)

    lam, lam_next = _lam(p0), _lam(p1)
    h = abs(lam_next - lam)
    hh = -h
    h_phi_1 = math.expm1(hh)
    b_h = h_phi_1

    rks: list[float] = []
    d1s: list = []
    for n in range(1, eo):
        q_n = preds.pop(0)
        rk = (_lam(_dpoints(sched, previous[-n].step)[0]) - lam) / h
        rks.append(rk if math.isfinite(rk) else 0)
        d1s.append((q_n - q) / rk)

    if prediction_next is not None:
        rks.append(1.0)
        order_check = 1
        d1s.append(prediction_next - q)
    else:
        order_check = 2

    if not rks or (eo == order_check and cfg["fast_solve"]):
        rhos = [0.5]
    else:
        h_phi_k = h_phi_1 / hh - 1
        rows, rhs = [], []
        for n in range(1, len(rks) + 1):
            rows.append([math.pow(v, n - 1) for v in rks])
            rhs.append(h_phi_k * math.factorial(n) / b_h)
            h_phi_k = h_phi_k / hh - 1 / math.factorial(n + 1)
        rhos = np.linalg.solve(rows, rhs).tolist()

    q = q + sumprod(rhos[: len(d1s)], d1s)
    return P.forward(space, cur.sample, q, p0, p1, cur.noise, eta=cfg["eta"])


def _stated(fn, cfg, cur: Rec, pred, sched, previous) -> Rec:
    "structured.py:106-125: result record repeats the *unmodified* input fields"
    return Rec(cur.sample, cur.prediction, cur.step, cur.noise, fn(cfg, cur, pred, sched, previous))


def unipc(cfg, cur: Rec, pred, sched: Sched, previous=()) -> Rec:
    "structured.py:469-497: convert -> UniC on the previous record -> predictor on corrected sample"
    p0 = _dpoints(sched, cur.step)[0]
    if cfg["deriv"]:
        cur = cur._replace(prediction=P.convert(pred, cfg["deriv"], cur.sample, cur.prediction, p0))
        pred = cfg["deriv"]
    same = bool(cfg["deriv"])  # from here on `pred` *is* the sampler's own derivative object
    if previous:
        corrected = unisolve(cfg, previous[-1], pred, sched, previous[:-1], prediction_next=cur.prediction, same=same)
        cur = cur._replace(sample=corrected)
    if cfg["predictor"]:
        return sample_packed(cfg["predictor"], cur, pred, sched, previous)
    return Rec(cur.sample, cur.prediction, cur.step, cur.noise, unisolve(cfg, cur, pred, sched, previous, same=same))


def spc(cfg, cur: Rec, pred, sched: Sched, previous=()) -> Rec:
    "structured.py:527-577: blend the input sample with a re-computed (corrected) previous step"
    p0 = _dpoints(sched, cur.step)[0]
    if cfg["deriv"]:
        cur = cur._replace(prediction=P.convert(pred, cfg["deriv"], cur.sample, cur.prediction, p0))
        pred = cfg["deriv"]
    if previous:
        shifted = [r._replace(prediction=q) for r, q in zip(previous, (*(r.prediction for r in previous[1:]), cur.prediction))]
        corrected = sample_packed(cfg["corrector"], shifted.pop(), pred, sched, shifted).final
        wp, wc = (p0.s, p0.a) if cfg["adaptive"] else (0, 0)
        wp, wc = softmax2(wp - cfg["bias"], wc + cfg["bias"])
        if cfg["invert"]:
            wp, wc = wc, wp
        if abs(cfg["power"] - 1) > 1e-8:
            mixed = spowf(spowf(cur.sample, cfg["power"]) * wp + spowf(corrected, cfg["power"]) * wc, 1 / cfg["power"])
        else:
            mixed = cur.sample * wp + corrected * wc
        cur = cur._replace(sample=mixed)
    return sample_packed(cfg["predictor"], cur, pred, sched, previous)


_STATED = {"euler": euler, "dpm": dpm, "adams": adams, "unip": unisolve}


def sample_packed(cfg: dict, cur: Rec, pred, sched: Sched, previous: Sequence[Rec] = ()) -> Rec:
    "structured.py:62-68 dispatch"
    k = cfg["kind"]
    if k in _STATED:
        return _stated(_STATED[k], cfg, cur, pred, sched, previous)
    if k == "unipc":
        return unipc(cfg, cur, pred, sched, previous)
    if k == "spc":
        return spc(cfg, cur, pred, sched, previous)
    raise KeyError(k)


def sample(cfg, sample_, prediction, step, pred, sched, noise=None, previous=()) -> Rec:
    "structured.py:70-86"
    return sample_packed(cfg, Rec(sample_, prediction, Stp(*step), noise), pred, sched, previous)


def adapter_loop(cfg, x, model, pred, sched: Sched, steps: int, include=slice(None), rng=None, callback=None):
    "interface.py:23-59: the canonical denoise loop for structured samplers"
    previous: list[Rec] = []
    points = sched.schedule(steps)
    keep = require_previous(cfg)
    for n, point in list(enumerate(points))[include]:
        step = stp_from_int(n, len(points))
        rec = sample_packed(
            cfg,
            Rec(x, model(x, *point), step, rng(step) if rng and require_noise(cfg) else None),
            pred,
            sched,
            previous,
        )
        if keep > 0:
            previous.append(rec)
            previous = previous[max(len(previous) - keep, 0) :]
        x = rec.final
        if callback:
            callback(x, n, (point, points[n + 1] if n + 1 < len(points) else (0, 0, 1)))
    return x


def generate(loop, x_init, rng, sched: Sched, steps: int, include=slice(None)):
    "functional.py:125-149 generate_model: draw (and scale) the initial sample, then run `loop(x)`"
    from .scalars import pt_add_noise

    if x_init is None and include.start is None:
        x = rng(None)
    else:
        x = pt_add_noise(sched.ipoint((include.start or 0) / steps), 0 if x_init is None else x_init, rng(None))
        x = x / pt_add_noise(sched.point(1), 0.0, 1.0)
    return loop(x)
