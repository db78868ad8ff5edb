"""rocFFT defect probe (GPU box): torch.fft ALONE, nothing of this package is imported.  400 random real transforms of small
shapes in one process, each checked against the CPU result.  On ROCm 7.2 / gfx950 about a dozen of them come back 20-60 % wrong
(multi-dimensional, power-of-two lengths, last axis 8 or 32), persistently for the plan concerned -- the reason skr_colored_any.hip
self-checks every new hipFFT plan on unit impulses and falls back to 1-D plans + its own direct-DFT kernels (Colored)."""
import random, torch
dev = torch.device("cuda:0")
rng = random.Random(2026)
torch.backends.cuda.cufft_plan_cache[0].max_size = int(__import__("os").environ.get("PLAN_CACHE", "1023"))
bad = 0
g = torch.Generator().manual_seed(0)
for i in range(400):
    nd = rng.choice((1, 2, 3))
    pow2 = rng.random() < 0.5
    dims = tuple(rng.choice((2, 4, 8, 16, 32)) if pow2 else rng.choice((3, 6, 12, 20, 24, 40)) for _ in range(nd))
    batch = rng.choice((1, 2, 3, 6, 16, 32, 48))
    x = torch.randn((batch, *dims), generator=g)
    d = tuple(range(-nd, 0))
    gpu = torch.fft.rfftn(x.to(dev), dim=d).cpu()
    cpu = torch.fft.rfftn(x, dim=d)
    err = (gpu - cpu).abs().max().item() / cpu.abs().max().item()
    back = torch.fft.irfftn(torch.fft.rfftn(x.to(dev), dim=d), s=dims, dim=d).cpu()
    err2 = (back - x).abs().max().item() / x.abs().max().item()
    if err > 1e-4 or err2 > 1e-4:
        bad += 1
        print("case", i, "batch", batch, "dims", dims, "forward err", err, "round-trip err", err2)
print("done, bad:", bad)
