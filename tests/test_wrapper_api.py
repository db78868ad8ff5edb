"""The scheduler protocol around step() against numbers recorded from the reference itself
(tests/golden/wrapper_api.json, written by tools/make_golden.py::wrapper_api from /root/reference/skrample/diffusers.py):
set_timesteps in its four calling forms (:494-538), timesteps / sigmas / init_noise_sigma / order / config (:256-279, :486-488),
add_noise / scale_noise / scale_model_input / time_shift / set_begin_index on CPU tensors (:375-388, :540-548), the diffusers-config
round trip (parse_diffusers_config :112-196, from_diffusers_config :417-462, as_diffusers_config :206-231) and the functional
bridge (:286-310).  Host-resident tensors: this runs without a GPU, through the package's host executor."""

import dataclasses
import json
import os

import numpy as np
import pytest
import torch
from conftest import GOLDEN, eq_nan

import skrample_amd.diffusers as PD
import skrample_amd.scheduling as PS
from skrample_amd.sampling import models as PM
from skrample_amd.sampling import structured as PT


@pytest.fixture(scope="module")
def api():
    return json.load(open(os.path.join(GOLDEN, "wrapper_api.json")))


MAKE = {
    "euler_scaled": lambda: PD.SkrampleWrapperScheduler(PT.Euler(), PS.Scaled()),
    "dpm2_karras": lambda: PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())),
    "unipc_flowshift": lambda: PD.SkrampleWrapperScheduler(PT.UniPC(order=3), PS.FlowShift(PS.Linear()), PM.FlowModel()),
    "adams_zsnr_v": lambda: PD.SkrampleWrapperScheduler(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()),
    "euler_beta_flowshift": lambda: PD.SkrampleWrapperScheduler(PT.Euler(), PS.FlowShift(PS.Beta(PS.ZSNR()))),
    "euler_exp_static": lambda: PD.SkrampleWrapperScheduler(PT.Euler(), PS.Exponential(PS.Scaled()), allow_dynamic=False),
    "rku3_scaled": lambda: PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3),
    "dyn3_linear": lambda: PD.DynasauRKWrapperScheduler(PS.Linear(), sampler_order=3, model=PM.FlowModel()),
}
FORMS = {
    "n7": dict(num_inference_steps=7),
    "n1": dict(num_inference_steps=1),
    "timesteps5": dict(timesteps=[900, 700, 500, 300, 100]),
    "sigmas4": dict(sigmas=[1.0, 0.7, 0.4, 0.1]),
    "n6_mu": dict(num_inference_steps=6, mu=0.8),
    "none": dict(),
}


def norm(v):
    "the fixture's rendering of config values (tools/make_golden.py::_norm)"
    if isinstance(v, type):
        return f"<{v.__name__}>"
    if isinstance(v, dict):
        return {str(k): norm(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [norm(x) for x in v]
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, torch.dtype):
        return str(v)
    return repr(v)


def close(got, ref, what, rtol=1e-12):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=1e-300, equal_nan=True, err_msg=what)


def same_config(got: dict, ref: dict, what: str) -> None:
    assert set(got) == set(ref), (what, sorted(set(got) ^ set(ref)))
    for k, r in ref.items():
        g = got[k]
        if isinstance(r, float) and isinstance(g, (int, float)):
            assert g == pytest.approx(r, rel=1e-12), (what, k)
        else:
            assert g == r, (what, k, g, r)


@pytest.mark.parametrize("name", MAKE)
def test_set_timesteps_call_forms(name, api):
    for form, kw in FORMS.items():
        ref = api["timesteps"][f"{name}/{form}"]
        w = MAKE[name]()
        w.set_timesteps(7)
        if "error" in ref:
            with pytest.raises(Exception) as info:
                w.set_timesteps(**kw)
                _ = w.timesteps
            assert type(info.value).__name__ == ref["error"], (name, form)
            continue
        w.set_timesteps(**kw)
        what = f"{name}/{form}"
        close(w.timesteps.tolist(), ref["timesteps"], what + " timesteps")
        close(w.sigmas.tolist(), ref["sigmas"], what + " sigmas")
        close(w.schedule_np.tolist(), ref["schedule_np"], what + " schedule_np")
        assert float(w.init_noise_sigma) == ref["init_noise_sigma"] and int(w.order) == ref["order"], what
        assert repr(w.schedule) == ref["schedule"], what  # dynamic Karras/Exponential steps and the mu shift land in the schedule
        same_config(norm(dict(w.config)), ref["config"], what + " config")


@pytest.mark.parametrize("name", MAKE)
def test_noise_and_input_scaling(name, api):
    ref = api["scale"][name]
    x = torch.tensor(api["x"], dtype=torch.float64).reshape(2, 3, 4, 4)
    nz = torch.tensor(api["noise"], dtype=torch.float64).reshape(2, 3, 4, 4)
    w = MAKE[name]()
    w.set_timesteps(6)
    ts = w.timesteps
    close([float(w.time_shift(0.7, 1.3, torch.tensor(t, dtype=torch.float64))) for t in (0.1, 0.5, 0.9)], ref["time_shift"], name + " time_shift")
    for k in (0, 2, len(ts) - 1):
        close(torch.as_tensor(w.scale_noise(x, ts[k], nz)).flatten().tolist(), ref[f"scale_noise/{k}"], f"{name} scale_noise {k}")
        close(torch.as_tensor(w.scale_model_input(x, ts[k])).flatten().tolist(), ref[f"scale_model_input/{k}"], f"{name} scale_model_input {k}")
        close(torch.as_tensor(w.scale_model_input(x, float(ts[k]))).flatten().tolist(), ref[f"scale_model_input_float/{k}"], f"{name} scale_model_input(float) {k}")
        close(torch.as_tensor(w.add_noise(x, nz, ts[k : k + 2])).flatten().tolist(), ref[f"add_noise/{k}"], f"{name} add_noise {k}")
    close(torch.as_tensor(w.add_noise(x, nz, ts[:0])).flatten().tolist(), ref["add_noise/empty"], name + " add_noise empty")
    if "begin_index/error" in ref:
        with pytest.raises(Exception) as info:
            w.set_begin_index(2 * w.order)
        assert type(info.value).__name__ == ref["begin_index/error"]
    else:
        w.set_begin_index(2 * w.order)
        assert dict(w.config).get("begin_index") == ref["begin_index/config"]
        close(torch.as_tensor(w.add_noise(x, nz, ts[2 * w.order : 2 * w.order + 1])).flatten().tolist(), ref["begin_index/add_noise"], name + " add_noise after set_begin_index")


def test_diffusers_config_round_trip(api):
    for key, ref in api["configs"].items():
        cfg = ref["config"]
        if key.startswith("override_"):
            kw = {"override_sampler": dict(sampler=PT.Adams), "override_schedule": dict(schedule=PS.Linear)}[key]
            parsed = PD.parse_diffusers_config(cfg, **kw)
            assert norm({f.name: getattr(parsed, f.name) for f in dataclasses.fields(parsed)}) == ref["parsed"], key
            continue
        if "error" in ref:
            with pytest.raises(Exception) as info:
                PD.SkrampleWrapperScheduler.from_diffusers_config(cfg)
            assert type(info.value).__name__ == ref["error"], key
            continue
        parsed = PD.parse_diffusers_config(cfg)
        assert norm({f.name: getattr(parsed, f.name) for f in dataclasses.fields(parsed)}) == ref["parsed"], key
        w = PD.SkrampleWrapperScheduler.from_diffusers_config(cfg)
        got = {"sampler": repr(w.sampler), "schedule": repr(w.schedule), "model": repr(w.model), "invert": bool(w.invert_prediction)}
        assert got == ref["wrapper"], key
        same_config(norm(PD.as_diffusers_config(w.sampler, w.schedule, w.model)), ref["as_config"], key + " as_diffusers_config")
        w.set_timesteps(5)
        close(w.timesteps.tolist(), ref["timesteps"], key + " timesteps")
        close(w.sigmas.tolist(), ref["sigmas"], key + " sigmas")


@pytest.mark.parametrize("name", ["euler_scaled", "dpm2_karras", "adams_zsnr_v", "rku3_scaled"])
def test_functional_bridge(name, api):
    ref = api["functional"][name]
    x = torch.tensor(api["x"], dtype=torch.float64).reshape(2, 3, 4, 4)
    draws = [torch.tensor(d, dtype=torch.float64).reshape(2, 3, 4, 4) for d in ref["draws"]]

    def toy(xx, t, s, a):
        return xx * 0.3 - 0.1 * s + 0.05 * a

    w = MAKE[name]()
    pool = list(draws)
    res = w.functional_sample_model(x.clone(), toy, 5, rng=lambda *_: pool.pop(0))
    assert len(draws) - len(pool) == ref["used"]
    close(torch.as_tensor(res).flatten().tolist(), ref["sample_model"], name + " functional_sample_model", rtol=1e-10)
    pool = list(draws)
    gen = w.functional_generate_model(toy, lambda *_: pool.pop(0), 5)
    assert len(draws) - len(pool) == ref["used_generate"]
    close(torch.as_tensor(gen).flatten().tolist(), ref["generate_model"], name + " functional_generate_model", rtol=1e-10)
    assert eq_nan([0.0], [0.0])
