"""ctypes binding of libskrample_hip.so (include/skrample_hip.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback anywhere in this
package.  If the library is missing or a launch fails, `SkrampleHipError` is raised.
"""

from __future__ import annotations

import ctypes
import os
import sys
import threading

import torch

MAX_TERMS = 80
ABI_VERSION = 13
SKR_ERR_UNSUPPORTED = 7  # include/skrample_hip.h: valid request outside what the fast kernels cover

BF16, F16, F32, F64, NONE = 0, 1, 2, 3, -1
DTYPE_CODE = {torch.bfloat16: BF16, torch.float16: F16, torch.float32: F32, torch.float64: F64}
CODE_DTYPE = {v: k for k, v in DTYPE_CODE.items()}

LIB_NAME = "libskrample_hip.so"
# (SKR_HIP_LIB: a differently built library for A/B measurements -- tools/ only; the product loads the in-tree one)
LIB_PATH = os.environ.get("SKR_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", LIB_NAME)

EXPORTS = (
    "skr_step_launch",
    "skr_step_launch_indexed",
    "skr_program_create",
    "skr_program_launch",
    "skr_program_destroy",
    "skr_tape_launch",
    "skr_noise_random",
    "skr_noise_offset",
    "skr_noise_brownian",
    "skr_noise_pyramid",
    "skr_noise_pyramid_any",
    "skr_noise_pyramid_nd",
    "skr_noise_colored",
    "skr_noise_colored_any",
    "skr_colorize",
    "skr_error_mean",
    "skr_power_blend",
    "skr_philox_u32",
    "skr_abi_version",
    "skr_strerror",
    "skr_last_hip_error",
    "skr_build_info",
    "skr_set_tuning",
    "skr_stat",
)


class SkrampleHipError(RuntimeError):
    "The HIP engine is unavailable or rejected a request.  Never silently degraded."


class StepPlanC(ctypes.Structure):
    "mirror of `skr_step_plan`"

    _fields_ = [
        ("n_terms", ctypes.c_int32),
        ("n_group_a", ctypes.c_int32),
        ("dtype_a", ctypes.c_int32),
        ("dtype_b", ctypes.c_int32),
        ("out0_dtype", ctypes.c_int32),
        ("out1_dtype", ctypes.c_int32),
        ("acc_f64", ctypes.c_int32),
        ("noise_mode", ctypes.c_int32),
        ("coef0", ctypes.c_double * MAX_TERMS),
        ("coef1", ctypes.c_double * MAX_TERMS),
        ("chain", ctypes.c_double),
        ("zeta0", ctypes.c_double),
        ("zeta1", ctypes.c_double),
        ("stream0", ctypes.c_uint64),
        ("stream1", ctypes.c_uint64),
        ("sample_numel", ctypes.c_int64),
        ("convert_to", ctypes.c_int32),
        ("convert_from", ctypes.c_int32),
        ("convert_k", ctypes.c_double * 4),
    ]


TAPE_MAX_OPS, TAPE_REGS, TAPE_MAX_INPUTS, TAPE_MAX_OUTPUTS = 96, 16, 24, 4  # include/skrample_hip.h SKR_TAPE_*
(TAPE_LOAD, TAPE_STORE, TAPE_MUL_S, TAPE_DIV_S, TAPE_ADD_S, TAPE_RSUB_S, TAPE_RDIV_S, TAPE_ADD, TAPE_SUB, TAPE_MUL, TAPE_DIV, TAPE_NEG,
 TAPE_ADD_MS, TAPE_SUB_MS, TAPE_RSUB_MS, TAPE_MULZ_S) = range(16)  # fmt: skip


class TapeOpC(ctypes.Structure):
    "mirror of `skr_tape_op`"

    _fields_ = [("code", ctypes.c_int32), ("dst", ctypes.c_int32), ("a", ctypes.c_int32), ("b", ctypes.c_int32), ("k", ctypes.c_double)]


class TapeC(ctypes.Structure):
    "mirror of `skr_tape`"

    _fields_ = [("n_ops", ctypes.c_int32), ("n_inputs", ctypes.c_int32), ("n_outputs", ctypes.c_int32), ("dtype", ctypes.c_int32), ("ops", TapeOpC * TAPE_MAX_OPS)]


ROW_TERMS = 16  # include/skrample_hip.h SKR_ROW_TERMS


class StepRowC(ctypes.Structure):
    "mirror of `skr_step_row`: the scalars of one launch, resident on the device for indexed launches"

    _fields_ = [
        ("coef0", ctypes.c_double * ROW_TERMS),
        ("coef1", ctypes.c_double * ROW_TERMS),
        ("chain", ctypes.c_double),
        ("zeta0", ctypes.c_double),
        ("zeta1", ctypes.c_double),
        ("stream0", ctypes.c_uint64),
        ("stream1", ctypes.c_uint64),
        ("convert_k", ctypes.c_double * 4),
    ]


def plan_structure(plan: StepPlanC) -> tuple:
    "what a captured launch freezes: everything of a plan except the scalars a row carries"
    return (plan.n_terms, plan.n_group_a, plan.dtype_a, plan.dtype_b, plan.out0_dtype, plan.out1_dtype, plan.acc_f64, plan.noise_mode,
            plan.sample_numel, plan.convert_to, plan.convert_from)  # fmt: skip


class IndexedRows:
    """Device-resident step scalars of a captured sampling loop (skr_step_launch_indexed).

    mode "record": launches run normally and append one row each (their structure is remembered);
    mode "emit"  : launches become indexed launches reading row `base + k` (k = position in the loop) -- used under graph capture;
    mode "refill": launches run normally (on whatever tensors the dry run uses) and overwrite row `slot*length + k` after checking
                   that the structure is the captured one -- how a captured loop is re-targeted to another schedule."""

    def __init__(self, device: torch.device, slots: int = 4):
        self.device, self.slots = device, slots
        self.mode, self.cursor, self.length = "record", 0, 0
        self.structures: list[tuple] = []
        self.host: list[StepRowC] = []
        self.rows_dev: torch.Tensor | None = None
        self.index_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.slot = 0

    @staticmethod
    def row_from(plan: StepPlanC) -> StepRowC:
        if plan.n_terms > ROW_TERMS:
            raise SkrampleHipError(f"a launch with {plan.n_terms} operands does not fit a device-resident row ({ROW_TERMS})")
        row = StepRowC()
        for k in range(plan.n_terms):
            row.coef0[k], row.coef1[k] = plan.coef0[k], plan.coef1[k]
        row.chain, row.zeta0, row.zeta1, row.stream0, row.stream1 = plan.chain, plan.zeta0, plan.zeta1, plan.stream0, plan.stream1
        for k in range(4):
            row.convert_k[k] = plan.convert_k[k]
        return row

    def finish_recording(self) -> None:
        self.length = len(self.host)
        size = ctypes.sizeof(StepRowC)
        self.rows_dev = torch.zeros(self.slots * self.length * size, dtype=torch.uint8, device=self.device)
        self.upload(0)

    def upload(self, slot: int) -> None:
        "copy the host rows of `slot` to the device (stream-ordered)"
        size = ctypes.sizeof(StepRowC)
        blob = b"".join(bytes(r) for r in self.host[slot * self.length : (slot + 1) * self.length])
        staging = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        self.rows_dev[slot * self.length * size : (slot + 1) * self.length * size].copy_(staging, non_blocking=False)

    def begin(self, mode: str, slot: int = 0) -> None:
        self.mode, self.cursor, self.slot = mode, 0, slot
        if mode == "refill":
            need = (slot + 1) * self.length
            while len(self.host) < need:
                self.host.append(StepRowC())

    def launch(self, lib, plan: StepPlanC, arr, out0_ptr, out1_ptr, seeds_ptr, numel: int, stream_ptr: int) -> int:
        k = self.cursor
        self.cursor += 1
        if self.mode == "record":
            self.structures.append(plan_structure(plan))
            self.host.append(self.row_from(plan))
            return lib.skr_step_launch(ctypes.byref(plan), arr, out0_ptr, out1_ptr, seeds_ptr, numel, stream_ptr)
        if k >= self.length:
            raise SkrampleHipError("more launches than the captured loop has")
        if self.mode == "refill":
            if plan_structure(plan)[:8] + plan_structure(plan)[9:] != self.structures[k][:8] + self.structures[k][9:]:
                raise SkrampleHipError(f"launch {k} of the new schedule has a different structure than the captured loop: re-capture")
            self.host[self.slot * self.length + k] = self.row_from(plan)
            return lib.skr_step_launch(ctypes.byref(plan), arr, out0_ptr, out1_ptr, seeds_ptr, numel, stream_ptr)
        if plan_structure(plan) != self.structures[k]:
            raise SkrampleHipError(f"launch {k} differs in structure between the recording pass and the capture")
        return lib.skr_step_launch_indexed(ctypes.byref(plan), arr, out0_ptr, out1_ptr, seeds_ptr, numel, self.rows_dev.data_ptr(), self.index_dev.data_ptr(), k, stream_ptr)


# Launch hooks, PER THREAD (a scheduler instance is single-threaded by contract, but several may run in different threads of one
# process, and a hook installed by one of them must not record or redirect the launches of another):
#   _hip.indexed  IndexedRows | None   installed by skrample_amd.graphs while it records / captures / re-targets a loop
#   _hip.trace    list | None          launches are appended while a wrapper lowers a step program (sampling/program.py)
# Both read and assign like plain module attributes (`_hip.trace = []`); the module's class routes them to thread-local storage.
_hooks = threading.local()


class _HipModule(type(sys)):
    @property
    def indexed(self) -> "IndexedRows | None":
        return getattr(_hooks, "indexed", None)

    @indexed.setter
    def indexed(self, value) -> None:
        _hooks.indexed = value

    @property
    def trace(self) -> "list | None":
        return getattr(_hooks, "trace", None)

    @trace.setter
    def trace(self, value) -> None:
        _hooks.trace = value


sys.modules[__name__].__class__ = _HipModule


def hooks_clear() -> bool:
    "no launch hook installed on this thread (the replayed-step fast path launches by program handle, which the hooks do not see)"
    return getattr(_hooks, "indexed", None) is None and getattr(_hooks, "trace", None) is None


def step_launch_raw(plan: StepPlanC, arr, out0_ptr, out1_ptr, seeds_ptr, numel: int, stream_ptr: int) -> int:
    "every skr_step_launch of the package goes through here (status returned, not checked)"
    lib = load()
    rows = getattr(_hooks, "indexed", None)
    if rows is not None:
        return rows.launch(lib, plan, arr, out0_ptr, out1_ptr, seeds_ptr, numel, stream_ptr)
    return lib.skr_step_launch(ctypes.byref(plan), arr, out0_ptr, out1_ptr, seeds_ptr, numel, stream_ptr)


_lock = threading.Lock()
_lib: ctypes.CDLL | None = None


def load() -> ctypes.CDLL:
    "dlopen the engine (once) and declare the prototypes"
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise SkrampleHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "skrample_amd has no CPU fallback."
            )
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as exc:  # pragma: no cover - depends on the box
            raise SkrampleHipError(f"cannot load {LIB_PATH}: {exc}") from exc
        vp, i64, u64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32
        lib.skr_step_launch.argtypes = [ctypes.POINTER(StepPlanC), ctypes.POINTER(vp), vp, vp, vp, i64, vp]
        lib.skr_step_launch.restype = ctypes.c_int
        lib.skr_step_launch_indexed.argtypes = [ctypes.POINTER(StepPlanC), ctypes.POINTER(vp), vp, vp, vp, i64, vp, vp, i32, vp]
        lib.skr_step_launch_indexed.restype = ctypes.c_int
        lib.skr_program_create.argtypes = [ctypes.POINTER(StepPlanC), i64, ctypes.POINTER(vp)]
        lib.skr_program_create.restype = ctypes.c_int
        lib.skr_program_launch.argtypes = [vp, ctypes.POINTER(vp), vp, vp, vp, u64, u64, vp]
        lib.skr_program_launch.restype = ctypes.c_int
        lib.skr_program_destroy.argtypes = [vp]
        lib.skr_program_destroy.restype = None
        lib.skr_tape_launch.argtypes = [ctypes.POINTER(TapeC), ctypes.POINTER(vp), ctypes.POINTER(vp), i64, vp]
        lib.skr_tape_launch.restype = ctypes.c_int
        lib.skr_noise_random.argtypes = [vp, i32, vp, u64, i64, i64, vp]
        lib.skr_noise_random.restype = ctypes.c_int
        lib.skr_noise_offset.argtypes = [vp, i32, vp, u64, u64, i64, ctypes.POINTER(i64), i32, ctypes.c_uint32, ctypes.c_double, vp]
        lib.skr_noise_offset.restype = ctypes.c_int
        lib.skr_noise_brownian.argtypes = [vp, i32, vp, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), i32, ctypes.c_double, vp, i32, i64, i64, vp]
        lib.skr_noise_brownian.restype = ctypes.c_int
        lib.skr_noise_pyramid.argtypes = [vp, i32, vp, vp, vp, vp, u64, u64, i64, i64, i64, i64, i32, ctypes.c_double, i32, i32, vp]
        lib.skr_noise_pyramid.restype = ctypes.c_int
        lib.skr_noise_pyramid_any.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, u64, u64, i64, i64, i64, i64, i32, ctypes.c_double, i32, i32, vp]
        lib.skr_noise_pyramid_any.restype = ctypes.c_int
        lib.skr_noise_pyramid_nd.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, u64, u64, i64, i32, ctypes.POINTER(i64), i32, i32, ctypes.c_double, i32, i32, vp]
        lib.skr_noise_pyramid_nd.restype = ctypes.c_int
        lib.skr_noise_colored.argtypes = [vp, i32, vp, vp, vp, i64, vp, u64, i64, i32, i32, i32, ctypes.c_double, i32, ctypes.c_double, vp]
        lib.skr_noise_colored.restype = ctypes.c_int
        lib.skr_noise_colored_any.argtypes = [vp, i32, vp, vp, vp, vp, u64, i64, i32, ctypes.POINTER(i32), ctypes.c_double, i32, ctypes.c_double, vp]
        lib.skr_noise_colored_any.restype = ctypes.c_int
        lib.skr_colorize.argtypes = [vp, i32, vp, vp, vp, i64, i32, ctypes.POINTER(i32), ctypes.c_double, i32, ctypes.c_double, vp]
        lib.skr_colorize.restype = ctypes.c_int
        lib.skr_error_mean.argtypes = [vp, vp, i32, i64, i32, vp, vp, vp]
        lib.skr_error_mean.restype = ctypes.c_int
        lib.skr_philox_u32.argtypes = [vp, u64, u64, u64, i64, vp]
        lib.skr_philox_u32.restype = ctypes.c_int
        lib.skr_power_blend.argtypes = [vp, i32, vp, i32, vp, i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, i64, vp]
        lib.skr_power_blend.restype = ctypes.c_int
        lib.skr_abi_version.restype = ctypes.c_int
        lib.skr_strerror.argtypes = [ctypes.c_int]
        lib.skr_strerror.restype = ctypes.c_char_p
        lib.skr_last_hip_error.restype = ctypes.c_int
        lib.skr_build_info.restype = ctypes.c_char_p
        lib.skr_stat.argtypes = [ctypes.c_char_p]
        lib.skr_stat.restype = ctypes.c_int64
        lib.skr_set_tuning.argtypes = [ctypes.c_char_p, i32]
        lib.skr_set_tuning.restype = ctypes.c_int
        if lib.skr_abi_version() != ABI_VERSION:
            raise SkrampleHipError(f"ABI mismatch: library {lib.skr_abi_version()} != binding {ABI_VERSION}; rebuild")
        _lib = lib
        return lib


def check(status: int, what: str) -> None:
    if status != 0:
        lib = load()
        msg = lib.skr_strerror(status).decode()
        extra = f" (hipError {lib.skr_last_hip_error()})" if status == 6 else ""
        raise SkrampleHipError(f"{what}: {msg}{extra}")


def require_device(t: torch.Tensor, what: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SkrampleHipError(
            f"{what} must be a torch tensor on a HIP device; skrample_amd executes on the GPU only "
            "(no CPU path). Got " + (f"{t.device}" if isinstance(t, torch.Tensor) else type(t).__name__)
        )


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the handle without building a torch.cuda.Stream object (1.5 -> 0.3 us per step)


def current_stream_ptr(device: torch.device) -> int:
    "hipStream_t of torch's current stream on `device`"
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


# When a list is installed as `_hip.trace` (per thread, see _HipModule) every launch appends
# (plan, inputs, out0, out1, seeds, numel): used by the step programs and by bench.py to lift the exact plans the
# samplers emit and replay them through the C ABI.


def launch_step(plan: StepPlanC, inputs: list[torch.Tensor], out0, out1, seeds, numel: int, device: torch.device) -> None:
    "one fused kernel launch on torch's current stream of `device`"
    load()
    trace = getattr(_hooks, "trace", None)
    if trace is not None:
        trace.append((plan, list(inputs), out0, out1, seeds, numel))
    n = len(inputs)
    arr = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in inputs])
    status = step_launch_raw(
        plan,
        arr,
        out0.data_ptr() if out0 is not None else None,
        out1.data_ptr() if out1 is not None else None,
        seeds.data_ptr() if seeds is not None else None,
        numel,
        current_stream_ptr(device),
    )
    check(status, "skr_step_launch")
