"""GPU box: randomised runs of the wrappers with noise drawn ahead on the side stream (prefetch_noise) against the same runs with
it off -- results must be bit-identical whatever the sampler, generator, shape, run length, restarts in mid-run and out-of-order
timesteps (usage: python tests/soak_ahead.py [first_seed last_seed])."""
import os, random, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests")]
import torch
import skrample_amd.diffusers as PD, skrample_amd.scheduling as PS
from skrample_amd.pytorch import noise as PN
from skrample_amd.sampling import structured as PT
from skrample_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
bad = raised = 0
for seed in range(first, last):
    rng = random.Random(seed)
    kind = rng.choice((PN.Pyramid, PN.Offset, PN.Colored))
    props = {PN.Pyramid: PN.PyramidProps(), PN.Offset: PN.OffsetProps(), PN.Colored: PN.ColoredProps()}[kind]
    which = rng.choice(("dpm2", "euler", "unipc3", "rk2", "rk3", "dyn2"))
    shape = (rng.choice((1, 2, 3)), rng.choice((1, 4)), rng.choice((16, 32)), rng.choice((16, 32, 64)))
    steps = rng.randint(3, 8)
    restart_at = rng.choice((None, None, 1, 2))
    jump = which in ("dpm2", "euler") and rng.random() < 0.3

    def make(prefetch):
        kw = dict(noise_type=kind, noise_props=props, prefetch_noise=prefetch)
        if which == "dpm2": return PD.SkrampleWrapperScheduler(PT.DPM(order=2, stochasticity=1), PS.Karras(PS.Scaled()), **kw)
        if which == "euler": return PD.SkrampleWrapperScheduler(PT.Euler(stochasticity=0.6), PS.Scaled(), **kw)
        if which == "unipc3": return PD.SkrampleWrapperScheduler(PT.UniPC(order=3, stochasticity=1), PS.Scaled(), **kw)
        if which == "rk2": return PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=2, stochasticity=1, **kw)
        if which == "rk3": return PD.RKUltraWrapperScheduler(PS.Scaled(), sampler_order=3, stochasticity=0.5, **kw)
        return PD.DynasauRKWrapperScheduler(PS.Scaled(), sampler_order=2, stochasticity=1, **kw)

    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(shape, generator=g).bfloat16().to(dev)
    pool = [torch.randn(shape, generator=g).bfloat16().to(dev) for _ in range(64)]
    seeds = [rng.randrange(2**40) for _ in range(shape[0])]

    def run(prefetch):
        w = make(prefetch)
        w._AHEAD_KINDS = (PN.Pyramid, PN.Offset, PN.Colored)
        outs, k = [], 0
        for attempt in range(2 if restart_at is not None else 1):
            w.set_timesteps(steps)
            ts = list(w.timesteps.tolist())
            order = list(range(len(ts)))
            if jump and len(order) > 3:
                order[1], order[2] = order[2], order[1]
            if attempt == 0 and restart_at is not None:
                order = order[: restart_at * getattr(w, "order", 1)]
            x = x0
            for i in order:
                x = w.step(pool[k % 64], ts[i], x, generator=seeds, return_dict=False)[0]
                k += 1
                outs.append(x.clone())
        torch.cuda.synchronize()
        return outs

    try:
        a, b = run(True), run(False)
        ok = len(a) == len(b) and all(torch.equal(u, v) for u, v in zip(a, b))
    except Exception as e:  # both settings must fail alike (e.g. a singular multistep history after an out-of-order step)
        try:
            run(False)
            ok = False
            print("seed", seed, "only the prefetching run failed:", type(e).__name__, str(e)[:200])
        except Exception:
            ok = True
            raised += 1
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, kind.__name__, which, shape, steps, restart_at, jump)
print(f"done, failures: {bad}  (runs that raise alike with and without drawing ahead: {raised} of {last - first})")
