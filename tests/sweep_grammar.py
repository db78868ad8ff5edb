"""The seeded grammar of wrapper configurations behind tests/golden/steps_sweep.npz (tools/make_golden.py::sweep), the live sweeps of
tools/sweep_vs_reference.py and the GPU box's device-vs-host soak (tests/soak_sweep.py).  A case is the TEXT of its constructor in a neutral
vocabulary -- W = diffusers module, T = structured samplers, S = schedules, M = models -- so that the reference, the package and the
tests build their objects from the same words.  Changing a line here changes the fixture: regenerate it."""


def sweep_spec(rng) -> tuple[str, str, tuple[int, ...], int]:
    eta = lambda: rng.choice(("0", "0", "0.5", "1", "-1.5", "0.3"))  # noqa: E731

    def deriv():
        return rng.choice(("", "", "", ", derivative_transform=None", ", derivative_transform=M.VelocityModel()", ", derivative_transform=M.FlowModel()"))

    def single(allow_unipc=True):
        kind = rng.choice(("Euler", "DPM", "DPM", "Adams", "Adams", "UniP", "UniPC", "UniPC") if allow_unipc else ("Euler", "DPM", "Adams", "UniP"))
        if kind == "Euler":
            return f"T.Euler(stochasticity={eta()})"
        if kind == "DPM":
            return f"T.DPM(order={rng.randint(1, 3)}, stochasticity={eta()}{deriv()})"
        if kind == "Adams":
            return f"T.Adams(order={rng.randint(1, 9)}, stochasticity={eta()}{deriv()})"
        fast = rng.choice(("", "", ", fast_solve=True"))
        if kind == "UniP":
            return f"T.UniP(order={rng.randint(1, 9)}, stochasticity={eta()}{fast}{deriv()})"
        pred = rng.choice(("", "", f", predictor={single(False)}"))
        return f"T.UniPC(order={rng.randint(1, 6)}, stochasticity={eta()}{fast}{pred}{deriv()})"

    flow = rng.random() < 0.4
    if flow:
        base = rng.choice(("S.Linear()", "S.Linear()", "S.Linear(sigma_start=0.9)"))
        mod = rng.choice(("{}", "{}", "S.FlowShift({})", "S.FlowShift({}, shift=1.7)", "S.Sinner({})", "S.Probit({})", "S.Beta({})", "S.Hyper({})", "S.FlowShift(S.Beta({}))"))
        model = rng.choice(("M.FlowModel()", "M.FlowModel()", "M.DataModel()", "M.VelocityModel()"))
    else:
        base = rng.choice(("S.Scaled()", "S.Scaled()", "S.ZSNR()", "S.Scaled(beta_scale=1)", "S.Scaled(beta_start=0.0001, beta_end=0.02, beta_scale=1)", "S.Scaled(base_timesteps=-1000, beta_scale=1)"))
        mod = rng.choice(("{}", "{}", "S.Karras({})", "S.Karras({}, rho=5.0)", "S.Exponential({})", "S.Beta({})", "S.Hyper({})", "S.Hyper({}, scale=3, tail=False)", "S.Sinner({})", "S.Hyper(S.Karras({}))"))
        model = rng.choice(("M.DataModel()", "M.VelocityModel()") if "ZSNR" in base else ("M.NoiseModel()", "M.NoiseModel()", "M.DataModel()", "M.VelocityModel()", "M.ScaleX()"))
    schedule = mod.format(base)
    opts = rng.choice(("", "", "", ", invert_prediction=True", ", compute_scale=torch.float64"))
    roll = rng.random()
    if roll < 0.12:
        text = f"W.RKUltraWrapperScheduler({schedule}, sampler_order={rng.randint(1, 6)}, stochasticity={eta()}, model={model}{opts})"
    elif roll < 0.2:
        text = f"W.DynasauRKWrapperScheduler({schedule}, sampler_order={rng.randint(2, 4)}, stochasticity={eta()}, model={model}{opts})"
    elif roll < 0.32:
        spc = f"T.SPC(predictor={single(False)}, corrector={single(False)}, bias={rng.choice(('0', '0.3', '-0.2'))}, power={rng.choice(('1', '2', '0.5'))}, adaptive={rng.choice(('True', 'False'))}, invert={rng.choice(('True', 'False'))})"
        text = f"W.SkrampleWrapperScheduler({spc}, {schedule}, {model}{opts})"
    else:
        text = f"W.SkrampleWrapperScheduler({single()}, {schedule}, {model}{opts})"
    dtype = rng.choice(("float32", "float32", "bfloat16", "float16"))
    if "float64" in opts and rng.random() < 0.5:
        dtype = "float64"
    shape = (rng.randint(1, 3), rng.randint(1, 4), rng.choice((4, 7, 8)), rng.choice((5, 8, 6)))
    steps_n = rng.randint(2, 9) if "RK" not in text else rng.randint(1, 3)
    return text, dtype, shape, steps_n


def native_spec(rng) -> tuple[str, str, tuple[int, ...], int]:
    """the same draw re-targeted at `compute_scale=None` on 16-bit tensors: the reference then computes in the TENSOR dtype, one rounded op at a
    time (tests/golden/steps_sweep_native.npz; results are compared bit for bit).  SPC with a signed-power blend (power != 1) is left out: this
    package evaluates that blend fused."""
    while True:
        text, dtype, shape, steps_n = sweep_spec(rng)
        if "T.SPC(" not in text or "power=1," in text:
            break
    text = text.replace(", compute_scale=torch.float64", "")[:-1] + ", compute_scale=None)"
    if dtype in ("float32", "float64"):
        dtype = rng.choice(("bfloat16", "float16"))
    return text, dtype, shape, steps_n


def functional_spec(rng):
    "a functional sampler (RKUltra, DynasauRK, adaptive RKMoire, the structured adapter), schedule, model, run length and step window -- texts over F / I / T / S / M"
    eta = lambda: rng.choice(("0", "0", "0.5", "1", "-1.5"))  # noqa: E731
    deriv = lambda: rng.choice(("", "", ", derivative_transform=None", ", derivative_transform=M.VelocityModel()", ", derivative_transform=M.FlowModel()"))  # noqa: E731
    roll = rng.random()
    if roll < 0.3:
        text = f"F.RKUltra(order={rng.randint(1, 9)}, stochasticity={eta()}{deriv()})"
    elif roll < 0.5:
        opts = rng.choice(("", ", invert=True", ", per_step_decay=0.1, total_step_decay=-0.02", ", per_step_decay=0.0", ", total_step_decay=0.3"))
        text = f"F.DynasauRK(order={rng.randint(2, 4)}, stochasticity={eta()}{opts}{deriv()})"
    elif roll < 0.8:
        opts = "".join(
            rng.sample(
                (", threshold=1e-3", ", initial=1 / 20", ", maximum=1 / 3", ", adaption=0.5", ", discard=2.0", ", rescale_init=False", ", rescale_max=True", ", evaluator=F.FunctionalAdaptive.mae"),
                rng.randint(0, 3),
            )
        )
        text = f"F.RKMoire(order={rng.randint(2, 7)}{opts}{deriv()})"
    else:
        kind = rng.choice(("T.Euler(stochasticity={e})", "T.DPM(order={o3}, stochasticity={e})", "T.Adams(order={o9})", "T.UniP(order={o9})", "T.UniPC(order={o6}, stochasticity={e})", "T.SPC()"))
        text = "I.StructuredFunctionalAdapter(" + kind.format(e=eta(), o3=rng.randint(1, 3), o9=rng.randint(1, 9), o6=rng.randint(1, 6)) + ")"
    if rng.random() < 0.4:
        schedule = rng.choice(("S.Linear()", "S.FlowShift(S.Linear())", "S.Sinner(S.Linear())", "S.Beta(S.Linear())"))
        model = rng.choice(("M.FlowModel()", "M.DataModel()", "M.VelocityModel()"))
    else:
        schedule = rng.choice(("S.Scaled()", "S.ZSNR()", "S.Karras(S.Scaled())", "S.Exponential(S.Scaled())", "S.Hyper(S.Scaled())", "S.Scaled(beta_scale=1)"))
        model = rng.choice(("M.DataModel()", "M.VelocityModel()") if "ZSNR" in schedule else ("M.NoiseModel()", "M.DataModel()", "M.VelocityModel()", "M.ScaleX()"))
    steps_n = rng.randint(1, 9)
    lo = rng.choice((None, None, 0, 1, 2))
    hi = rng.choice((None, None, steps_n, max(steps_n - 1, 1), 5))
    return text, schedule, model, steps_n, (lo, hi)
