"""The functional samplers against trajectories recorded from the reference itself (tests/golden/functional_api.json, written by
tools/make_golden.py::functional_api): RKUltra (functional.py:218-270), DynasauRK (:272-351), the adaptive RKMoire (:353-472) and
StructuredFunctionalAdapter (interface.py:14-59) drive a toy model over CPU float64 tensors.  Checked: the result of sample_model
and generate_model, every (timestep, sigma, alpha) the model was evaluated at (so RKMoire's accepted / rejected step sizes), the
callback trace, and how many draws were consumed.  Host-resident tensors: runs without a GPU through the host executor."""

import json
import os

import numpy as np
import pytest
import torch
from cases import MODELS, SCHEDULES
from conftest import GOLDEN

from skrample_amd.sampling import functional as F
from skrample_amd.sampling import interface as I  # noqa: E741
from skrample_amd.sampling import structured as S

API = json.load(open(os.path.join(GOLDEN, "functional_api.json")))


def close(got, ref, what, rtol=1e-9):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=1e-12, err_msg=what)


def run_case(name, dev):
    ref = API["cases"][name]
    sampler = eval(ref["expr"], {"F": F, "S": S, "I": I})  # the same constructor text the fixture was recorded with
    sched, model_t = SCHEDULES[ref["schedule"]][1](), MODELS[ref["model"]][1]
    steps, (lo, hi) = ref["steps"], ref["include"]
    x = torch.tensor(API["x"], dtype=torch.float64).reshape(2, 3, 4).to(dev)
    if ref["adjust_steps"] is not None:
        assert sampler.adjust_steps(steps) == ref["adjust_steps"]
    seen, trace = [], []

    def toy(xx, t, s, a):
        seen.append([float(t), float(s), float(a)])
        xx = torch.as_tensor(xx)
        return xx * 0.3 - 0.1 * s + 0.05 * a + 0.01 * torch.sin(xx * 3.0)

    def cb(sample, n, dp):
        trace.append([int(n), *[float(v) for v in dp.point_from], *[float(v) for v in dp.point_to], float(torch.as_tensor(sample).double().sum())])

    if "error" in ref:
        with pytest.raises(Exception) as info:
            sampler.sample_model(x.clone(), toy, model_t, sched, steps, slice(lo, hi), lambda *_: x, cb)
        assert type(info.value).__name__ == ref["error"]
        return
    pool = [torch.tensor(d, dtype=torch.float64).reshape(2, 3, 4).to(dev) for d in ref["draws"]]
    n_draws = len(pool)
    res = sampler.sample_model(x.clone(), toy, model_t, sched, steps, slice(lo, hi), lambda *_: pool.pop(0), cb)
    assert n_draws - len(pool) == ref["used"] and not pool, name
    close(seen, ref["seen"], name + ": points the model was evaluated at")
    close(trace, ref["trace"], name + ": callback trace", rtol=1e-8)
    close(torch.as_tensor(res).cpu().flatten().tolist(), ref["result"], name + ": sample_model")
    seen.clear()
    pool = [torch.tensor(d, dtype=torch.float64).reshape(2, 3, 4).to(dev) for d in ref["generate_draws"]]
    gen = sampler.generate_model(toy, model_t, sched, lambda *_: pool.pop(0), steps, slice(lo, hi), None if lo is None else x.clone())
    assert not pool and len(seen) == ref["generate_nfe"], name
    close(torch.as_tensor(gen).cpu().flatten().tolist(), ref["generate"], name + ": generate_model")


@pytest.mark.parametrize("name", API["cases"])
def test_functional_sampler_trajectories(name):
    run_case(name, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", API["cases"])
def test_functional_sampler_trajectories_on_device(name):
    "the same recorded trajectories with device-resident float64 tensors: every stage is one fused HIP launch (fp64 accumulators)"
    from skrample_amd import _hip

    _hip.load()
    run_case(name, torch.device("cuda:0"))
