"""The sampler-level API in the reference's own arithmetic (skrample_amd/sampling/native.py + csrc/skr_tape.hip).

Called on tensors directly, the reference's samplers run one rounded torch op per arithmetic operation in the TENSOR dtype
(structured.py:70-86, 167-497).  native.py records that sequence; skr_tape_launch replays it in one launch.  Checked here:
  * no GPU: the RECORDER -- the tape of every step of the twelve reference-recorded 16-bit runs (tests/golden/native16.npz), interpreted
    with plain torch ops on CPU (one op = one torch call, as the reference), returns the reference's bits; so do tapes of random
    sampler / model / schedule combinations against the oracle's native-dtype chain, in bf16, fp16 and fp32;
  * GPU (test_step_gpu.py / below): the KERNEL -- StructuredSampler.sample on device tensors equals the same fixtures bit for bit."""

import zlib

import pytest
import torch
from cases import MODELS, NATIVE16_ORACLE, NATIVE16_TAGS, SAMPLERS, SCHEDULES, native16_case
from conftest import load_npz

from skr_oracle import samplers as OA
from skrample_amd import _hip
from skrample_amd.common import Step
from skrample_amd.sampling import native
from skrample_amd.sampling import structured as PT


def interpret(tape: native.Tape, results):
    "the tape, one torch op per entry, on the tape's own (CPU) leaves: what separate aten kernels in the tensor dtype compute"
    vals = []
    for code, a, b, k in tape.ops:
        if code == _hip.TAPE_LOAD:
            vals.append(tape.leaves[a])
        elif code == _hip.TAPE_MUL_S:
            vals.append(vals[a] * k)
        elif code == _hip.TAPE_DIV_S:
            vals.append(vals[a] / k)
        elif code == _hip.TAPE_ADD_S:
            vals.append(vals[a] + k)
        elif code == _hip.TAPE_RSUB_S:
            vals.append(k - vals[a])
        elif code == _hip.TAPE_RDIV_S:
            vals.append(vals[a].reciprocal() if k == 1.0 else (torch.tensor(k, dtype=torch.float64 if tape.dtype == torch.float64 else torch.float32) / vals[a].to(torch.float64 if tape.dtype == torch.float64 else torch.float32)).to(tape.dtype))
        elif code == _hip.TAPE_ADD:
            vals.append(vals[a] + vals[b])
        elif code == _hip.TAPE_SUB:
            vals.append(vals[a] - vals[b])
        elif code == _hip.TAPE_MUL:
            vals.append(vals[a] * vals[b])
        elif code == _hip.TAPE_DIV:
            vals.append(vals[a] / vals[b])
        elif code == _hip.TAPE_NEG:
            vals.append(-vals[a])
        else:
            raise AssertionError(code)
        assert vals[-1].dtype == tape.dtype
    return [vals[v.n] for v in results]


def launch_form(tape: native.Tape, results):
    """The skr_tape native._allocate builds for the launch, run on the CPU as the kernel runs it: a file of SKR_TAPE_REGS registers, every operand
    number under skr_tape_launch's own checks, reads of defined registers only.  Returns the tensors `results` name."""
    ops, used_leaves, stores = native._allocate(tape, results)
    assert len(ops) <= _hip.TAPE_MAX_OPS and len(used_leaves) <= _hip.TAPE_MAX_INPUTS and 1 <= len(stores) <= _hip.TAPE_MAX_OUTPUTS
    file, outs = {}, {}
    scalar = {_hip.TAPE_MUL_S: lambda x, k: x * k, _hip.TAPE_DIV_S: lambda x, k: x / k, _hip.TAPE_ADD_S: lambda x, k: x + k, _hip.TAPE_RSUB_S: lambda x, k: k - x,
              _hip.TAPE_RDIV_S: lambda x, k: x.reciprocal(), _hip.TAPE_NEG: lambda x, k: -x, _hip.TAPE_MULZ_S: lambda x, k: x * k + 0}
    binary = {_hip.TAPE_ADD: lambda x, y, k: x + y, _hip.TAPE_SUB: lambda x, y, k: x - y, _hip.TAPE_MUL: lambda x, y, k: x * y, _hip.TAPE_DIV: lambda x, y, k: x / y,
              _hip.TAPE_ADD_MS: lambda x, y, k: x + y * k, _hip.TAPE_SUB_MS: lambda x, y, k: x - y * k, _hip.TAPE_RSUB_MS: lambda x, y, k: y * k - x}
    for code, dst, a, b, k in ops:
        assert 0 <= dst < _hip.TAPE_REGS
        if code == _hip.TAPE_LOAD:
            assert 0 <= a < len(used_leaves)
            file[dst] = tape.leaves[used_leaves[a]]
        elif code == _hip.TAPE_STORE:
            assert 0 <= b < len(stores) and b not in outs
            outs[b] = file[a]
        else:
            assert code != _hip.TAPE_RDIV_S or k == 1.0
            file[dst] = scalar[code](file[a], k) if code in scalar else binary[code](file[a], file[b], k)
    return [tape.leaves[tape.ops[v.n][1]] if tape.ops[v.n][0] == _hip.TAPE_LOAD else outs[stores[v.n]] for v in results]


def interpret_both(tape: native.Tape, results):
    "the op list as recorded, and the launch form of it (registers allocated): same tensors, bit for bit"
    plain = interpret(tape, results)
    for want, got in zip(plain, launch_form(tape, results)):
        assert got.dtype == want.dtype and torch.equal(torch.nan_to_num(got.double()), torch.nan_to_num(want.double())) and torch.equal(torch.isnan(got), torch.isnan(want))
    return plain


def record(sampler, x, out, step, model, sched, noise, previous):
    packed = PT.SampleInput(x, out, step, noise)
    if type(sampler) is PT.UniPC:
        tape, res = native.record_unipc(sampler, packed, model, sched, previous, require_device=False)
        s, p, f = interpret_both(tape, res)
        return PT.SKSamples(s, p, step, noise, f), tape
    tape, res = native.record_stated(sampler, packed, model, sched, previous, require_device=False)
    return PT.SKSamples(x, out, step, noise, interpret_both(tape, res)[0]), tape


@pytest.mark.parametrize("tag", NATIVE16_TAGS)
def test_recorded_steps_return_the_reference_bits(tag):
    dt, steps, mname, sname, expr, t = native16_case(load_npz("native16.npz"), tag)
    sampler, sched, model = eval(expr, {"S": PT}), SCHEDULES[sname][1](), MODELS[mname][1]
    previous = []
    longest = 0
    for i in range(steps):
        x, out, nz = t["x"][i], t["out"][i], t["noise"][i]
        rec, tape = record(sampler, x, out, Step.from_int(i, steps), model, sched, nz if sampler.require_noise else None, tuple(previous))
        assert rec.final.dtype == dt and torch.equal(rec.final, t["final"][i]), (tag, i)
        assert torch.equal(torch.as_tensor(rec.prediction), t["prediction"][i]), (tag, i)
        longest = max(longest, len(tape.ops))
        previous.append(rec)
        keep = sampler.require_previous
        previous = previous[max(len(previous) - keep, 0) :] if keep else []
    assert longest <= _hip.TAPE_MAX_OPS


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("name", ["euler_sde", "dpm1_sde", "dpm2_sde", "dpm3_sde", "adams4_sde", "adams9", "unip2_fast", "unip4_sde", "unipc1", "unipc2_fast", "unipc3_sde",
                                  "unipc2_adams3", "dpm2_deriv_v", "unipc3_deriv_flow_sde", "adams3_noderiv", "unipc3_noderiv"])
def test_recorded_steps_equal_the_oracles_native_chain(name, dtype):
    "every sampler family x model x schedule: the recorded tape == the oracle's reference-order chain run in the tensor dtype, bit for bit"
    mk_o, mk_p = SAMPLERS[name]
    steps, shape = 9, (2, 3, 8, 8)
    for sname, mname in (("karras_scaled", "eps"), ("linear", "flow"), ("zsnr", "v"), ("scaled", "scalex"), ("scaled", "data")):
        g = torch.Generator().manual_seed(zlib.crc32(f"{name}/{sname}/{mname}".encode()))
        cfg, sampler = mk_o(), mk_p()
        sched, osched, (omodel, model) = SCHEDULES[sname][1](), SCHEDULES[sname][0](), MODELS[mname]
        x = torch.randn(shape, generator=g).to(dtype)
        previous, oprevious = [], []
        for i in range(steps - (1 if sname == "zsnr" else 0)):
            out = torch.randn(shape, generator=g).to(dtype)
            nz = torch.randn(shape, generator=g).to(dtype)
            noise = nz if sampler.require_noise else None
            rec, tape = record(sampler, x, out, Step.from_int(i, steps), model, sched, noise, tuple(previous))
            ref = OA.sample(cfg, x, out, (i / steps, (i + 1) / steps), omodel, osched, noise, oprevious)
            assert ref.final.dtype == dtype
            same = torch.equal(rec.final, ref.final) or (torch.isnan(ref.final).any() and torch.equal(torch.isnan(rec.final), torch.isnan(ref.final)))
            assert same, (name, sname, mname, i, (rec.final.float() - ref.final.float()).abs().max())
            previous.append(rec)
            oprevious.append(ref)
            keep = sampler.require_previous
            previous = previous[max(len(previous) - keep, 0) :] if keep else []
            oprevious = oprevious[max(len(oprevious) - keep, 0) :] if keep else []
            x = ref.final


@pytest.mark.parametrize("tag", NATIVE16_TAGS)
def test_host_tensors_take_the_same_path(tag):
    "StructuredSampler.sample on host-resident bf16 / fp16 tensors == the reference-recorded runs, bit for bit (same dispatch as on the device)"
    dt, steps, mname, sname, expr, t = native16_case(load_npz("native16.npz"), tag)
    sampler, sched, model = eval(expr, {"S": PT}), SCHEDULES[sname][1](), MODELS[mname][1]
    previous = []
    for i in range(steps):
        rec = sampler.sample(t["x"][i], t["out"][i], Step.from_int(i, steps), model, sched, t["noise"][i] if sampler.require_noise else None, tuple(previous))
        assert torch.equal(rec.final, t["final"][i]) and torch.equal(torch.as_tensor(rec.prediction), t["prediction"][i]), (tag, i)
        previous.append(rec)
        keep = sampler.require_previous
        previous = previous[max(len(previous) - keep, 0) :] if keep else []


def test_register_allocation_and_refusals():
    "the launch form of a tape: registers below SKR_TAPE_REGS, every read of a live value, results stored once; CPU tensors never launch"
    sampler, sched, model = PT.DPM(order=3, stochasticity=1), SCHEDULES["scaled"][1](), MODELS["eps"][1]
    g = torch.Generator().manual_seed(3)
    shape, steps = (2, 4, 8, 8), 9
    mk = lambda: torch.randn(shape, generator=g).bfloat16()  # noqa: E731
    previous = [PT.SKSamples(mk(), mk(), Step.from_int(i, steps), None, mk()) for i in (2, 3)]
    packed = PT.SampleInput(mk(), mk(), Step.from_int(4, steps), mk())
    tape, res = native.record_stated(sampler, packed, model, sched, previous, require_device=False)
    assert sum(1 for op in tape.ops if op[0] == _hip.TAPE_LOAD) == 7  # x, out, noise, two history pairs
    launched = native.launches
    rec = native.try_stated(sampler, packed, model, sched, previous)  # host tensors: the tape op by op in torch, never a launch
    assert rec is not None and native.launches == launched and torch.equal(rec.final, interpret(tape, res)[0])
    with pytest.raises(native._Refused):
        native.record_stated(sampler, PT.SampleInput(packed.sample, packed.prediction.float(), packed.step, None), model, sched, previous, require_device=False)
    with pytest.raises(native._Refused):
        native.record_stated(PT.SPC(), packed, model, sched, previous, require_device=False)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_add_noise_and_remove_noise_round_as_the_reference_does(dtype):
    """Point.add_noise / remove_noise (reference common.py:32-40) are tensor expressions too: `sample * alpha + noise * sigma` is three
    rounded ops on 16-bit tensors.  What scheduler.add_noise / scale_noise hand a pipeline (img2img start) equals those ops bit for bit."""
    import skrample_amd.diffusers as PD
    import skrample_amd.scheduling as PS
    from skrample_amd.common import Point
    from skrample_amd.sampling import structured as PT

    g = torch.Generator().manual_seed(9)
    x, n = torch.randn(3, 4, 9, 7, generator=g).to(dtype), torch.randn(3, 4, 9, 7, generator=g).to(dtype)
    pt = Point(613.0, 0.7391, 0.6733)
    before = native.launches
    assert torch.equal(pt.add_noise(x, n), x * pt.alpha + n * pt.sigma)
    assert torch.equal(pt.remove_noise(x, n), (x - n * pt.sigma) / pt.alpha)
    assert native.launches == before  # (host tensors: the tape runs one torch op per entry)
    w = PD.SkrampleWrapperScheduler(PT.DPM(order=2), PS.Karras(PS.Scaled()))
    w.set_timesteps(7)
    t, _, sigma, alpha = w.timesteps[3], *w.schedule_np[3]
    want = x * float(alpha) + n * float(sigma)
    assert torch.equal(w.scale_noise(x, t, n), want) and torch.equal(w.add_noise(x, n, w.timesteps[3:4]), want)
    rk = PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3)
    rk.set_timesteps(4)
    _, sigma, alpha = rk.schedule_np[5]
    assert torch.equal(rk.scale_noise(x, rk.timesteps[5], n), x * float(alpha) + n * float(sigma))
    # "never": the fused form, one rounding -- within half a unit of the exact value, not the reference's bits
    keep, native.mode = native.mode, "never"
    try:
        fused = pt.add_noise(x, n)
    finally:
        native.mode = keep
    exact = x.double() * pt.alpha + n.double() * pt.sigma
    assert (fused.double() - exact).abs().max() <= (pt.add_noise(x, n).double() - exact).abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float64])
def test_remove_noise_at_alpha_zero_divides_as_the_reference_does(dtype):
    "a tensor over alpha = 0 is inf / nan in the reference (only its float path catches ZeroDivisionError, common.py:37-40): the same tensor here"
    from skrample_amd.common import Point

    g = torch.Generator().manual_seed(10)
    x, n = torch.randn(2, 5, generator=g).to(dtype), torch.randn(2, 5, generator=g).to(dtype)
    x[0, 0] = 0.0
    n[0, 0] = 0.0
    pt = Point(1000.0, 1.0, 0.0)
    want = (x - n * pt.sigma) / pt.alpha
    got = pt.remove_noise(x, n)
    assert got.dtype == dtype and torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))
    assert torch.isnan(got[0, 0]) and torch.isinf(got[0, 1])
    assert pt.remove_noise(0.5, 2.0) == 2.0  # the float path: the scaled noise


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_functional_samplers_on_16_bit_tensors_follow_the_generic_arithmetic(dtype):
    """functional.step_tableau on a 16-bit tensor sample (reference functional.py:55-108: generic tensor operators, every one rounded): each stage input and
    each weighted result is one recorded expression; on host tensors the tape runs one torch op per entry, so the result equals the expression written out
    with tensor operators.  (tools/sweep_vs_reference.py `functional` compares RKUltra / DynasauRK / the adapter with the imported reference bit for bit.)"""
    import math

    import skrample_amd.scheduling as PS
    from skrample_amd.common import DeltaPoint
    from skrample_amd.sampling import functional as PF
    from skrample_amd.sampling import models as PM
    from skrample_amd.sampling import tableaux

    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 5, generator=g).to(dtype)
    net = lambda xx, t, s, a: xx * 0.3 - 0.1 * s + 0.05 * a  # noqa: E731
    schedule, model = PS.Scaled(), PM.NoiseModel()
    tab = PF.RKUltra(order=2).tableau()
    step = Step(0.25, 0.5)
    (got,) = PF.step_tableau(tab, x, net, model, schedule, step)
    nodes, weights = tab
    s0, s1, *fr = schedule.ipoints([step[0], step[1], *(step[0] + c * (step[1] - step[0]) for c, _ in nodes)])
    delta = DeltaPoint(s0, s1)
    ds = []
    for frac, (_c, row) in zip(fr, nodes):
        if row:
            mix = 0
            for d, q in zip(ds, row):
                mix = mix + d * q
            d2 = DeltaPoint(s0, frac)
            xin = 0 + x * model.gamma(d2, 0) + (mix / math.fsum(row)) * model.delta(d2, 0)
        else:
            xin = x
        ds.append(net(xin, *frac))
    mix = 0
    for d, q in zip(ds, weights):
        mix = mix + d * q
    want = 0 + x * model.gamma(delta, 0) + mix * model.delta(delta, 0)
    assert got.dtype == dtype and torch.equal(got, want)
    assert tableaux  # (imported for the providers' side effects in some builds)
    # a full loop keeps the dtype, and differs from the fused form (native.mode = "never") by the chain's own rounding only
    chained = PF.RKUltra(order=4).sample_model(x, net, model, schedule, 5)
    keep, native.mode = native.mode, "never"
    try:
        fused = PF.RKUltra(order=4).sample_model(x, net, model, schedule, 5)
    finally:
        native.mode = keep
    assert chained.dtype == fused.dtype == dtype and (chained.float() - fused.float()).abs().max() <= 0.05 * fused.float().abs().max()
