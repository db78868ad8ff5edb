"""GPU box: run the seeded random parity sweeps of tests/test_step_gpu.py over many more seeds than the test-suite does
(usage: python tools/soak.py [first_seed last_seed]).  Found the fp16 fused multiply-convert double-rounding mismatch."""
import sys, os, traceback
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import torch
import test_step_gpu as T
dev = torch.device("cuda:0")
from skrample_amd import _hip; _hip.load()
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (48, 700)
for seed in range(lo, hi):
    for fn in (T.test_random_sweep_vs_oracle, T.test_random_sweep_runge_kutta_vs_oracle):
        try:
            fn.__wrapped__(seed, dev) if hasattr(fn, "__wrapped__") else fn(seed, dev)
        except Exception as e:
            bad += 1
            print("FAIL", fn.__name__, seed, type(e).__name__, str(e)[:300])
print("done, failures:", bad)
