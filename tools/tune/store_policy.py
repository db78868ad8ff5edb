"""Store cache policy of the streaming stores (skr_pack.h: `sc0 sc1` write-through) vs nt / plain / sc1, per launch shape.
Variant libraries are built by hand into tools/tune/libskrample_hip_<policy>.so; SKR_LIB picks one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from skrample_amd import _hip
if os.environ.get("SKR_LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["SKR_LIB"])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_plan as BP
S4 = 4 * 128 * 128
for k in (1, 2, 3, 4):
    BP.bench(f"K={k} bf16 -> bf16", 256, S4, k, 0, False, False, switches=[{}])
BP.bench("K=4 bf16 + philox", 256, S4, 4, 0, False, True, switches=[{}])
for k in (2, 3, 5):
    BP.bench(f"rk stage K={k}", 64, 4 * 256 * 256, k, 0, False, False, rk=True, switches=[{}])
BP.bench("two-out NA=8 NB=1 philox", 256, 16 * 128 * 128, 8, 1, True, True, switches=[{}])
