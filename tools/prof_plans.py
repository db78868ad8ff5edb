"""rocprofv3 target for the launch shapes below the headline (round 3): the two-output UniPC kernels and the Runge-Kutta stage
kernels through the C ABI on COLD rotating buffers (tools/bench_plan.py's harness, short runs).

  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_plans -o n --output-format csv -- python3 tools/prof_plans.py
  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/prof_plans.py        (one counter group per pass)
tools/summarize_counters.py turns the passes into profiles/r03_k2_counters.json / r03_rk1_counters.json."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_plan import S4, S16, bench

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "two"):
    bench("two-out NA=8 NB=1 philox", 256, S16, 8, 1, True, True, switches=[{}], iters=20)
    bench("two-out NA=10 NB=1", 256, S16, 10, 1, True, False, switches=[{}], iters=20)
if which in ("all", "rk"):
    for k in (2, 3, 4, 5, 7):
        bench(f"rk stage K={k}", 64, 4 * 256 * 256, k, 0, False, False, rk=True, switches=[{}], iters=40)
if which in ("all", "k"):
    for k in (1, 2, 4, 10, 14, 18):
        bench(f"K={k} bf16 -> bf16", 256, S4, k, 0, False, False, switches=[{}], iters=40)
    bench("K=4 bf16 -> bf16 + philox", 256, S4, 4, 0, False, True, switches=[{}], iters=40)
    bench("K=4 bf16 -> bf16 + philox B=64", 64, S4, 4, 0, False, True, switches=[{}], iters=40, footprint=1.0e9)
