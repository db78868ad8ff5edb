"""skrample_amd.common against the values the reference's own common.py produced (tests/golden/common_api.json, written by
tools/make_golden.py::common_api from /root/reference/skrample/common.py:24-213): floats, numpy arrays and CPU tensors, edge values
(zeros, infinities, negative bases, zero divisors) and the exceptions the reference raises included."""

import json
import math
import os

import numpy as np
import pytest
import torch
from conftest import GOLDEN

import skrample_amd.common as C

API = json.load(open(os.path.join(GOLDEN, "common_api.json")))


def num(v):
    if isinstance(v, (tuple, list)):
        return [num(x) for x in v]
    if isinstance(v, torch.Tensor):
        return num(v.tolist())
    if isinstance(v, np.ndarray):
        return num(v.tolist())
    v = float(v)
    return v if math.isfinite(v) else repr(v)


def attempt(fn):
    try:
        return num(fn())
    except Exception as exc:
        return {"error": type(exc).__name__}


def same(got, ref, what):
    if isinstance(ref, list):
        assert isinstance(got, list) and len(got) == len(ref), (what, got, ref)
        for i, (g, r) in enumerate(zip(got, ref)):
            same(g, r, f"{what}[{i}]")
    elif isinstance(ref, (dict, str)):
        assert got == ref, (what, got, ref)
    else:
        assert isinstance(got, float), (what, got, ref)
        assert got == pytest.approx(ref, rel=1e-14, abs=0) and math.copysign(1, got) == math.copysign(1, ref), (what, got, ref)


xs = [float(v) for v in API["xs"]]
arr = np.asarray(API["arr"])
ten = torch.tensor(arr)


def test_scalar_helpers():
    same([[attempt(lambda a=a, b=b: C.divf(a, b)) for b in (-2.0, -0.0, 0.0, 3.0)] for a in (-1.5, 0.0, 2.0)], API["divf"], "divf")
    same([attempt(lambda x=x: C.ln(x)) for x in xs], API["ln"], "ln")
    same([attempt(lambda x=x: C.rescale_positive(x)) for x in xs[:-1]], API["rescale_positive"], "rescale_positive")
    same([attempt(lambda x=x: C.rescale_subnormal(x)) for x in xs], API["rescale_subnormal"], "rescale_subnormal")
    same([attempt(lambda x=x: C.clamp(x)) for x in xs] + [attempt(lambda: C.clamp(5.0, 2, 3)), attempt(lambda: C.clamp(-5.0, -1, 3))], API["clamp"], "clamp")
    same([attempt(lambda n=n: C.bashforth(n)) for n in range(1, 10)], API["bashforth"], "bashforth")


def test_sample_generic_helpers():
    "the `Sample` generic: float, ndarray and tensor through the same code (common.py:11-17)"
    same([attempt(lambda x=x: C.exp(x)) for x in xs[:10]] + [attempt(lambda: C.exp(arr)), attempt(lambda: C.exp(ten))], API["exp"], "exp")
    same([attempt(lambda x=x: C.sigmoid(x)) for x in xs[:10]] + [attempt(lambda: C.sigmoid(arr)), attempt(lambda: C.sigmoid(ten))], API["sigmoid"], "sigmoid")
    same([attempt(lambda: C.softmax((0.1, -2.0, 3.0))), attempt(lambda: C.softmax((arr, arr * 0.5, arr - 1))), attempt(lambda: C.softmax((ten, ten * 2)))], API["softmax"], "softmax")
    got = [[attempt(lambda x=x, f=f: C.spowf(x, f)) for f in (0.5, 1.0, 2.0, -1.0)] for x in xs[:10]]
    got += [[attempt(lambda f=f: C.spowf(arr, f)), attempt(lambda f=f: C.spowf(ten, f))] for f in (0.5, 2.0)]
    same(got, API["spowf"], "spowf")
    same([attempt(lambda: C.mean(0.75)), attempt(lambda: C.mean(arr)), attempt(lambda: C.mean(ten))], API["mean"], "mean")
    same([attempt(lambda: C.normalize(0.3, 2.0)), attempt(lambda: C.normalize(arr, 4.0, 1.0)), attempt(lambda: C.normalize(ten, 0.5, -0.5))], API["normalize"], "normalize")
    same([attempt(lambda: C.regularize(0.3, 2.0)), attempt(lambda: C.regularize(arr, 4.0, 1.0)), attempt(lambda: C.regularize(ten, 0.5, -0.5))], API["regularize"], "regularize")


def test_points_steps_and_merge_strategies():
    p, p0, pz = C.Point(700.0, 1.3, 0.4), C.Point(0.0, 0.0, 1.0), C.Point(999.0, 1.0, 0.0)
    ref = API["point"]
    same([attempt(lambda: p.add_noise(0.5, -2.0)), attempt(lambda: p.add_noise(arr, arr[::-1].copy())), attempt(lambda: p.add_noise(ten, ten * 3))], ref["add_noise"], "Point.add_noise")
    got = [attempt(lambda: p.remove_noise(0.5, -2.0)), attempt(lambda: p.remove_noise(arr, arr[::-1].copy())), attempt(lambda: pz.remove_noise(0.5, -2.0)), attempt(lambda: p0.remove_noise(ten, ten * 3))]
    same(got, ref["remove_noise"], "Point.remove_noise")
    same(attempt(lambda: C.DeltaPoint(p, p0).difference()), ref["difference"], "DeltaPoint.difference")
    for rec in API["step"]:
        n, amount = rec["from_int"]
        st = C.Step.from_int(n, amount)
        what = f"Step.from_int({n}, {amount})"
        same(num(st), rec["step"], what)
        same(attempt(st.distance), rec["distance"], what + ".distance")
        same(attempt(st.position), rec["position"], what + ".position")
        same(attempt(st.amount), rec["amount"], what + ".amount")
        same([attempt(lambda k=k: st.offset(k)) for k in (-2, 0.5, 3)], rec["offset"], what + ".offset")
        same([attempt(lambda k=k: st.offset(k).clamp()) for k in (-9, 0, 9.5)], rec["clamp"], what + ".clamp")
        same(attempt(lambda: C.Step(st.time_to, st.time_from).normal()), rec["normal"], what + ".normal")
    a, b = list(range(0, 11)), list(range(0, 15, 2))
    assert {m.name for m in C.MergeStrategy} == set(API["merge"])
    for m in C.MergeStrategy:
        ref = API["merge"][m.name]
        assert str(m.value) == ref["value"]
        assert m.merge(a, b) == ref["ab"] and m.merge(b, a) == ref["ba"], m
        assert m.merge(a, b, lambda u, v: u // 2 == v // 2) == ref["cmp"], m
