"""Micro-benchmark of single launch plans through the C ABI under different kernel-selection switches
(skr_set_tuning), on rotating buffer sets larger than the Infinity Cache.

  python tools/bench_plan.py            # the standard list
Each line: plan, switches, us per launch (HIP events over 200 launches), algorithmic TB/s, fraction of 8 TB/s."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skrample_amd import _hip
from skrample_amd.sampling.lazy import empty_output

# Where the buffers sit.  "engine" (default): tensors the ENGINE allocates in a real run -- every output, and the inputs that are
# earlier engine results (samples, stored derivatives, the fp32 state: even operand slots here) -- come from its output allocator
# (lazy.empty_output: start shifted by 4 KiB + k * 8 KiB inside the allocation); network outputs (odd slots) come straight from
# torch.  "torch": everything straight from torch's allocator -- same-sized tensors then sit exact multiples of their size apart
# (32 MiB for the RK shard), the worst case for the HBM channel map (round 2's numbers were taken this way).
PLACEMENT = os.environ.get("SKR_BENCH_PLACEMENT", "engine")

dev = torch.device("cuda:0")
lib = _hip.load()
stream = torch.cuda.current_stream(dev).cuda_stream


def bench(name, batch, sample, n_a, n_b, two_out, noise, rk=False, dtype=torch.bfloat16, switches=(), iters=200, footprint=1.2e9):
    n = batch * sample
    item = 2 if dtype != torch.float32 else 4
    per_set = n * (n_a * item + n_b * 4 + (4 + item if two_out else item) + (item if rk else 0))
    nsets = max(2, min(8, int(footprint // per_set) + 1))
    g = torch.Generator(device=dev).manual_seed(1)
    sets = []
    def alloc(dt, engine_owned, fill=True):
        t = empty_output((n,), dt, dev) if (engine_owned and PLACEMENT == "engine") else torch.empty(n, device=dev, dtype=dt)
        if fill:
            t.copy_(torch.randn(n, device=dev, generator=g))
        return t

    for _ in range(nsets):
        ins = [alloc(dtype, k % 2 == 0) for k in range(n_a)] + [alloc(torch.float32, True) for _ in range(n_b)]
        o0 = alloc(torch.float32 if two_out else dtype, True, fill=False)
        o1 = alloc(dtype, True, fill=False) if (two_out or rk) else None
        sets.append((ins, o0, o1))
    seeds = torch.arange(batch, dtype=torch.int64, device=dev) + 42
    code = _hip.DTYPE_CODE[dtype]
    plan = _hip.StepPlanC()
    plan.n_terms, plan.n_group_a, plan.dtype_a, plan.dtype_b = n_a + n_b, n_a, code, _hip.F32 if n_b else code
    plan.out0_dtype = _hip.F32 if two_out else code
    plan.out1_dtype = code if (two_out or rk) else -1
    plan.sample_numel, plan.chain = sample, 0.5
    for k in range(n_a + n_b):
        plan.coef0[k], plan.coef1[k] = 0.1 * (k + 1), -0.05 * (k + 1)
    if noise:
        plan.noise_mode, plan.zeta0, plan.zeta1, plan.stream0, plan.stream1 = 1, 0.3, 0.2 if two_out else 0.0, 1, 2
    if rk:
        plan.convert_to, plan.convert_from = 1, 1
        for i, v in enumerate((0.7, 0.9, 0.4, 1.3)):
            plan.convert_k[i] = v
    calls = []
    for ins, o0, o1 in sets:
        ptrs = (ctypes.c_void_p * len(ins))(*[t.data_ptr() for t in ins])
        calls.append((ptrs, o0.data_ptr(), o1.data_ptr() if o1 is not None else None))
    bytes_per = per_set
    out = []
    for sw in switches or ({},):
        lib.skr_set_tuning(b"reset", 0)
        for k, v in sw.items():
            assert lib.skr_set_tuning(k.encode(), v) == 0
        def run(count):
            for i in range(count):
                ptrs, p0, p1 = calls[i % nsets]
                st = lib.skr_step_launch(ctypes.byref(plan), ptrs, p0, p1, seeds.data_ptr() if noise else None, n, stream)
                if st:
                    _hip.check(st, "skr_step_launch")
        best = None
        # conditioning: the launch time of a sustained run settles only after ~16 ms ("Launch time over a long run")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(20); e1.record(); torch.cuda.synchronize()
        run(int(30e3 / max(e0.elapsed_time(e1) * 1e3 / 20, 1.0)) + 1)
        for rep in range(3):
            run(20)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(iters); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            best = us if best is None else min(best, us)
        tbs = bytes_per / best / 1e6
        print(f"{name:44s} {str(sw):28s} {best:8.2f} us  {tbs:6.3f} TB/s  {tbs / 8:.3f}", flush=True)
    lib.skr_set_tuning(b"reset", 0)
    del sets


S4 = 4 * 128 * 128
S16 = 16 * 128 * 128
OLD, NEW = {"one_trip": 0, "two_out": 0}, {}
if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "b64":  # BASELINE config 2's own batch: 2048 workgroups, one resident wave of them
        sw = [OLD, NEW, {"pace": 0}] + [{"xmap": x} for x in (0, 3, 5, 6, 8)] + [{"xmap": x, "pace": 0} for x in (0, 5)]
        bench("K=4 bf16 -> bf16 + philox  B=64", 64, S4, 4, 0, False, True, switches=sw, iters=400, footprint=1.0e9)
        bench("K=4 bf16 -> bf16           B=64", 64, S4, 4, 0, False, False, switches=sw, iters=400, footprint=1.0e9)
    if which in ("all", "k"):
        for k in (1, 2, 3, 4, 5, 6, 8):
            bench(f"K={k} bf16 -> bf16", 256, S4, k, 0, False, False, switches=[OLD, NEW, {"pace": 0}])
        for k in (2, 4, 6):
            bench(f"K={k} bf16 -> bf16 + philox", 256, S4, k, 0, False, True, switches=[OLD, NEW, {"pace": 0}])
    if which in ("all", "bigk"):  # round 3: Adams-Bashforth 5-9 / UniP >= 5 (10-18 operands), UniPC / SPC of order 5-9
        for k in (10, 14, 18, 20):
            bench(f"K={k} bf16 -> bf16", 256, S4, k, 0, False, False, switches=[OLD, NEW])
        bench("K=12 bf16 -> bf16 + philox", 256, S4, 12, 0, False, True, switches=[OLD, NEW])
        for na in (12, 16, 20):
            bench(f"two-out NA={na} NB=1", 256, S4, na, 1, True, False, switches=[OLD, NEW])
        bench("two-out NA=14 NB=1 philox", 256, S4, 14, 1, True, True, switches=[OLD, NEW])
    if which in ("all", "two"):
        for na, nb, noise in ((8, 1, True), (8, 1, False), (10, 1, False), (6, 1, True), (4, 1, True), (4, 0, False)):
            bench(f"two-out NA={na} NB={nb} {'philox' if noise else ''}", 256, S16, na, nb, True, noise, switches=[OLD, NEW, {"pace": 0}])
    if which in ("all", "rk"):
        for k in (2, 3, 4, 5, 6, 7, 8):
            bench(f"rk stage K={k}", 64, 4 * 256 * 256, k, 0, False, False, rk=True, switches=[OLD, NEW])
    if which == "ab":  # round 4: the two findings of the round-3 harness, A/B/A/B through the library on one box
        for k in (2, 3, 4):
            bench(f"rk stage K={k}", 64, 4 * 256 * 256, k, 0, False, False, rk=True, switches=[{"rk_blk": 256}, {"rk_blk": 128}] * 3)
        for na in (8, 10):
            bench(f"two-out NA={na} NB=1 (no in-kernel noise)", 256, S16, na, 1, True, False, switches=[{"two_nt": 0}, {"two_nt": 1}] * 3)
    if which in ("all", "f32"):
        for k in (2, 4):
            bench(f"K={k} f32 -> f32 + philox", 256, S4, k, 0, False, True, dtype=torch.float32, switches=[OLD, NEW, {"pace": 0}])
            bench(f"K={k} f32 -> f32", 256, S4, k, 0, False, False, dtype=torch.float32, switches=[OLD, NEW, {"pace": 0}])
