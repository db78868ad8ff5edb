"""Print the launch plans (operand counts, dtypes, outputs, noise) the samplers emit per step -- used to decide which
compile-time kernel shapes are worth instantiating.  Needs a GPU (noise generators)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd import _hip
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.sampling import structured as PT, models as PM
from skrample_amd.pytorch import noise as PN

dev = torch.device("cuda:0")
NAMES = {0: "bf16", 1: "f16", 2: "f32", 3: "f64", -1: "-"}


def run(name, w, shape=(2, 16, 64, 64), steps=8, dtype=torch.bfloat16):
    w.set_timesteps(steps)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(shape, device=dev, generator=g).to(dtype)
    seen = {}
    _hip.trace = []
    try:
        for t in w.timesteps:
            for _ in range(getattr(w, "order", 1)):
                out = torch.randn(shape, device=dev, generator=g).to(dtype)
                _hip.trace.clear()
                x = w.step(out, t, x, generator=list(range(shape[0])), return_dict=False)[0]
                for plan, inputs, o0, o1, seeds, numel in _hip.trace:
                    key = (plan.n_terms, plan.n_group_a, NAMES[plan.dtype_a], NAMES[plan.dtype_b] if plan.n_group_a < plan.n_terms else "-", NAMES[plan.out0_dtype], NAMES[plan.out1_dtype],
                           "philox" if plan.noise_mode == 1 and (plan.zeta0 or plan.zeta1) else "-", (plan.convert_to, plan.convert_from))
                    seen[key] = seen.get(key, 0) + 1
    finally:
        _hip.trace = None
    print(name)
    for k, v in seen.items():
        print(f"    x{v:<3d} terms={k[0]} groupA={k[1]} {k[2]}/{k[3]} out0={k[4]} out1={k[5]} noise={k[6]} conv={k[7]}")


W = PD.SkrampleWrapperScheduler
run("dpm2 sde", W(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())))
run("dpm3 ode", W(PT.DPM(order=3), PS.Scaled()))
run("adams4 v zsnr", W(PT.Adams(order=4), PS.ZSNR(), PM.VelocityModel()))
for o in (2, 3, 4):
    run(f"unipc{o} sde flow philox", W(PT.UniPC(order=o, stochasticity=1), PS.Linear(), PM.FlowModel()))
    run(f"unipc{o} ode flow", W(PT.UniPC(order=o), PS.Linear(), PM.FlowModel()))
run("unipc3 sde flow colored", W(PT.UniPC(order=3, stochasticity=1), PS.Linear(), PM.FlowModel(), noise_type=PN.Colored, noise_props=PN.ColoredProps()))
run("unipc3 sde eps fp16", W(PT.UniPC(order=3, stochasticity=1), PS.Scaled()), dtype=torch.float16)
run("unip3 ode", W(PT.UniP(order=3), PS.Scaled()))
run("spc default", W(PT.SPC(), PS.Scaled()))
run("spc dpm2/adams2 sde", W(PT.SPC(predictor=PT.DPM(order=2, stochasticity=1), corrector=PT.Adams(order=2)), PS.Scaled()))
run("rk6 sde pyramid", PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=6, stochasticity=1, noise_type=PN.Pyramid, noise_props=PN.PyramidProps()), shape=(2, 4, 64, 64), steps=3)
run("rk4 ode", PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=4), shape=(2, 4, 64, 64), steps=3)
run("dpm2 sde fp32 latents", W(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled())), dtype=torch.float32)
