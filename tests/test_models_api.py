"""skrample_amd.sampling.models against values recorded from the reference's own models.py (tests/golden/models_api.json, written
by tools/make_golden.py::models_api from /root/reference/skrample/sampling/models.py:11-239): six model transforms x three
schedules x four step spans -- to_x / from_x, gamma / delta / zeta / zeta_ts / eta_transform at four eta values, forward / backward
with and without noise on CPU float64 tensors and on floats -- and ModelConvert between every ordered pair."""

import json
import math
import os

import pytest
import torch
from conftest import GOLDEN

import skrample_amd.scheduling as RS
from skrample_amd.common import DeltaPoint
from skrample_amd.sampling import models as M

API = json.load(open(os.path.join(GOLDEN, "models_api.json")))
X, O, NZ = (torch.tensor(API[k], dtype=torch.float64).reshape(2, 3) for k in ("x", "o", "noise"))
SCHEDS = {"scaled": RS.Scaled(), "linear": RS.Linear(), "zsnr": RS.ZSNR()}


def num(v):
    v = float(v)
    return v if math.isfinite(v) else repr(v)


def attempt(fn):
    try:
        r = fn()
        if isinstance(r, tuple) and not isinstance(r, torch.Tensor):
            return [[num(q) for q in p] for p in r]
        if isinstance(r, (int, float)):
            return num(r)
        return [num(v) for v in torch.as_tensor(r).flatten().tolist()]
    except Exception as exc:
        return {"error": type(exc).__name__}


def nonfinite_only(ref) -> bool:
    return isinstance(ref, list) and len(ref) > 0 and all(isinstance(v, str) or nonfinite_only(v) for v in ref)


def same(got, ref, what):
    if got == {"error": "ZeroDivisionError"} and nonfinite_only(ref):
        # stated difference (models.py::_reciprocal): a zero denominator (e.g. eps-prediction at alpha = 0) makes the reference's
        # TENSOR path fill the result with inf / nan silently, while its scalar path raises ZeroDivisionError; here both raise
        return
    if isinstance(ref, list):
        assert isinstance(got, list) and len(got) == len(ref), (what, got, ref)
        for i, (g, r) in enumerate(zip(got, ref)):
            same(g, r, f"{what}[{i}]")
    elif isinstance(ref, (dict, str)):
        assert got == ref, (what, got, ref)
    else:
        assert isinstance(got, float), (what, got, ref)
        assert got == pytest.approx(ref, rel=1e-11, abs=1e-13), (what, got, ref)


@pytest.mark.parametrize("key", API["cases"])
def test_model_transform(key):
    ref = API["cases"][key]
    expr, sname, a, b = key.split("|")
    m = eval(expr, {"M": M})
    assert repr(m) == ref["repr"]
    dp = DeltaPoint(*SCHEDS[sname].ipoints([float(a), float(b)]))
    same(attempt(lambda: m.to_x(X, O, dp.point_from)), ref["to_x"], key + " to_x")
    same(attempt(lambda: m.from_x(X, O, dp.point_from)), ref["from_x"], key + " from_x")
    same(attempt(lambda: m.to_x(0.7, -0.4, dp.point_from)), ref["to_x_float"], key + " to_x(float)")
    for eta in (0.0, 0.5, 1.0, -1.5):
        got = [attempt(lambda: m.gamma(dp, eta)), attempt(lambda: m.delta(dp, eta)), attempt(lambda: m.zeta(dp, eta)), attempt(lambda: m.zeta_ts(dp, eta))]
        same(got, ref[f"gdz/{eta}"], f"{key} gamma/delta/zeta eta={eta}")
        same(attempt(lambda: m.eta_transform(dp, eta)), ref[f"eta_transform/{eta}"], f"{key} eta_transform eta={eta}")
        same(attempt(lambda: m.forward(X, O, dp, NZ, eta)), ref[f"forward/{eta}"], f"{key} forward eta={eta}")
        same(attempt(lambda: m.backward(X, O, dp, NZ, eta)), ref[f"backward/{eta}"], f"{key} backward eta={eta}")
    same(attempt(lambda: m.forward(X, O, dp)), ref["forward/plain"], key + " forward")
    same(attempt(lambda: m.backward(X, O, dp)), ref["backward/plain"], key + " backward")
    same(attempt(lambda: m.forward(0.7, -0.4, dp, 0.2, 1.0)), ref["forward/float"], key + " forward(float)")


def test_model_convert_between_every_pair():
    pt = RS.Scaled().ipoint(0.4)
    for pair, ref in API["convert"].items():
        ea, eb = pair.split("->")
        cv = M.ModelConvert(eval(ea, {"M": M}), eval(eb, {"M": M}))
        wrapped = cv.wrap_model_call(lambda xx, t, s, a: xx * 0.3 - 0.1 * s + 0.05 * a)
        same(attempt(lambda: cv.output_to(X, O, pt)), ref["output_to"], pair + " output_to")
        same(attempt(lambda: cv.output_from(X, O, pt)), ref["output_from"], pair + " output_from")
        same(attempt(lambda: wrapped(X, *pt)), ref["wrapped"], pair + " wrap_model_call")
