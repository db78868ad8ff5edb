"""skrample_amd.scheduling against numbers recorded from the reference's own scheduling.py (tests/golden/scheduling_api.json,
written by tools/make_golden.py::scheduling_api from /root/reference/skrample/scheduling.py:23-664): 33 schedule expressions --
every base, sub-schedule and modifier with default and non-default parameters, nested stacks -- each with schedule_np at four
run lengths, points / ipoints, step / istep, the end points, the sigma space and the modifier-stack introspection."""

import json
import os

import numpy as np
import pytest
from conftest import GOLDEN

import skrample_amd.scheduling as R
from skrample_amd.common import Step

API = json.load(open(os.path.join(GOLDEN, "scheduling_api.json")))


def close(got, ref, what):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-13, equal_nan=True, err_msg=what)


@pytest.mark.parametrize("expr", API["cases"])
def test_schedule_against_the_reference(expr):
    ref = API["cases"][expr]
    sch = eval(expr, {"R": R})  # the constructor text the fixture was recorded with
    assert repr(sch) == ref["repr"] and type(sch.space).__name__ == ref["space"]
    ts = API["t"]
    for n in (1, 2, 9, 30):
        close(sch.schedule_np(n).tolist(), ref[f"schedule_np/{n}"], f"{expr} schedule_np({n})")
        close([list(p) for p in sch.schedule(n)], ref[f"schedule_np/{n}"], f"{expr} schedule({n})")
    close([list(p) for p in sch.points(ts)], ref["points"], expr + " points")
    close(sch.points_np(ts).tolist(), ref["points"], expr + " points_np")
    close([list(p) for p in sch.ipoints(ts)], ref["ipoints"], expr + " ipoints")
    close([list(sch.ipoint(t)) for t in ts], ref["ipoints"], expr + " ipoint")
    close(list(sch.point_0), ref["point_0"], expr + " point_0")
    close(list(sch.point_1), ref["point_1"], expr + " point_1")
    st = Step.from_int(2, 9)
    close([list(p) for p in sch.step(st)], ref["step"], expr + " step")
    close([list(p) for p in sch.istep(st)], ref["istep"], expr + " istep")
    if "all_split" in ref:
        mods, sub, base = sch.all_split
        assert [[repr(m) for m in mods], repr(sub), repr(base)] == ref["all_split"]
        assert repr(sch.lowest) == ref["lowest"] and [repr(v) for v in sch.all] == ref["all"]
        assert repr(sch.find(R.FlowShift)) == ref["find_flowshift"] and repr(sch.find(R.Hyper, exact=True)) == ref["find_hyper_exact"]
        found = sch.find_split(R.FlowShift)
        got = None if found is None else [[repr(m) for m in found[0]], repr(found[1]), [repr(m) for m in found[2]], repr(found[3]), repr(found[4])]
        assert got == ref["find_split_flowshift"]
        assert repr(sch.stack(mods, sub, base)) == ref["restacked"]


def test_fixed_schedule_and_sigma_spaces():
    ref = API["fixed"]
    fixed = R.FixedSchedule.from_regular(np.asarray([900.0, 600.0, 300.0, 50.0]), np.asarray([10.0, 3.0, 0.8, 0.05]), R.VariancePreserving())
    close(fixed.schedule_np(4).tolist(), ref["schedule_np/4"], "FixedSchedule.schedule_np")
    close([list(p) for p in fixed.points([0.0, 0.3, 1.0])], ref["points"], "FixedSchedule.points")
    for name, space in (("vp", R.VariancePreserving()), ("flow", R.FlowMatching())):
        r = API[f"space/{name}"]
        if "error" in r:
            with pytest.raises(Exception) as info:
                space.normalize(np.asarray([0.0, 0.05, 0.7, 1.0, 14.6]))
            assert type(info.value).__name__ == r["error"]
            continue
        close([np.asarray(a).tolist() for a in space.normalize(np.asarray([0.0, 0.05, 0.7, 1.0, 14.6]))], r["normalize"], name + " normalize")
        close(np.asarray(space.regularize(np.asarray([0.0, 0.05, 0.5, 0.9]))).tolist(), r["regularize"], name + " regularize")
    sch = R.Karras(R.Scaled())
    close(R.np_schedule_lru(sch, 6).tolist(), API["lru"]["np"], "np_schedule_lru")
    close([list(p) for p in R.schedule_lru(sch, 6)], API["lru"]["points"], "schedule_lru")
    assert R.schedule_lru(sch, 6) is R.schedule_lru(R.Karras(R.Scaled()), 6)  # cached on the (hashable) schedule value
