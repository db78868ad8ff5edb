"""Lazy linear forms over device tensors -- the bridge between the fp64 host math and the fused
HIP step kernel.

Every solver step of every sampler is a linear combination  sum_k c_k * T_k  of a handful of
same-shaped tensors (SURVEY.md "three facts" #2).  The samplers in this package therefore never
execute tensor arithmetic: they run their (scalar, fp64) algebra on `Lin` objects, which only
accumulate coefficients, and `evaluate()` turns the final one or two forms into ONE launch of
`skr_step_launch` (include/skrample_hip.h).  Plain Python numbers flow through the same code
unchanged -- that is host scalar logic (schedule dry-runs, coefficient tests).

Operand residency decides where a form is evaluated, never availability of the library:
  * HIP-device tensors  -> the fused kernel, always (a missing libskrample_hip.so raises; nothing falls back);
  * host-resident operands (CPU torch tensors, numpy arrays -- the reference's generic `T`, common.py:11-17, and
    BASELINE config 1 "on CPU torch") -> `_host_evaluate`, the same  sum_k c_k * T_k  in plain torch on the host.
    Device tensors can never reach it (mixed residency is refused).
"""

from __future__ import annotations

import contextvars
import math
import os
from typing import Any, Sequence

import torch

from .. import _hip
from .._hip import SkrampleHipError

NUMBER = (int, float)

# dtype in which *derived, persisted* state is kept (UniPC's corrected sample) and, when float64,
# the accumulator precision.  The scheduler wrapper sets it from `compute_scale`.
_compute_dtype: contextvars.ContextVar = contextvars.ContextVar("skr_compute_dtype", default=None)


class compute_scale:
    "context manager: `with compute_scale(torch.float32): sampler.sample_packed(...)`"

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.token = _compute_dtype.set(self.dtype)
        return self

    def __exit__(self, *exc):
        _compute_dtype.reset(self.token)


class PhiloxNoise:
    """Symbolic standard-normal tensor N(seeds[b], stream) of a given shape: drawn inside the step
    kernel (0 bytes of HBM traffic), or realised into a tensor on demand.
    Element e of sample b comes from Philox4x32-10 block e//4, lane e%4, key = seeds[b],
    counter high words = stream (oracle/skr_oracle/noise.py::philox_normal)."""

    __slots__ = ("seeds", "stream", "shape", "device", "_cache")

    def __init__(self, seeds: torch.Tensor, stream: int, shape: Sequence[int], device: torch.device):
        self.seeds, self.stream, self.shape, self.device = seeds, int(stream), tuple(shape), device
        self._cache: dict | None = None

    @classmethod
    def quick(cls, seeds: torch.Tensor, stream: int, shape: tuple, device: torch.device) -> "PhiloxNoise":
        "the same object without argument conversion (the replayed-step fast path makes one per step)"
        self = cls.__new__(cls)
        self.seeds, self.stream, self.shape, self.device, self._cache = seeds, stream, shape, device, None
        return self

    @property
    def sample_numel(self) -> int:
        return math.prod(self.shape[1:]) if len(self.shape) > 1 else (self.shape[0] if self.shape else 1)

    def numel(self) -> int:
        return math.prod(self.shape)

    def realize(self, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        "materialise through skr_noise_random (any shape)"
        if self._cache is None:
            self._cache = {}
        if dtype not in self._cache:
            out = torch.empty(self.shape, dtype=dtype, device=self.device)
            lib = _hip.load()
            batch = self.shape[0] if len(self.shape) > 1 else 1
            _hip.check(
                lib.skr_noise_random(out.data_ptr(), _hip.DTYPE_CODE[dtype], self.seeds.data_ptr(), self.stream, batch, self.numel() // max(batch, 1), _hip.current_stream_ptr(self.device)),
                "skr_noise_random",
            )
            self._cache[dtype] = out
        return self._cache[dtype]

    def fusable(self) -> bool:
        return len(self.shape) > 1 and self.sample_numel % 8 == 0 and self.numel() > 0


class Node:
    "leaf standing for the value of another form that is stored as the first output of the same launch"

    __slots__ = ("form",)

    def __init__(self, form):
        self.form = form


class RoundedConversion:
    """out0 of a launch defined not as a linear form but as the reference's *op-by-op rounded*
    conversion from_x(s, to_x(s, o)) in the tensors' own dtype (include/skrample_hip.h, convert_*).
    Used by the Runge-Kutta wrapper, where the reference converts before widening to compute_scale."""

    def __init__(self, sample: torch.Tensor, output: torch.Tensor, to_kind: int, from_kind: int, k: Sequence[float]):
        _check_tensor(sample), _check_tensor(output)
        if sample.dtype != output.dtype or sample.shape != output.shape:
            raise SkrampleHipError("rounded conversion needs sample and output of one dtype and shape")
        self.sample, self.output, self.to_kind, self.from_kind, self.k = sample, output, int(to_kind), int(from_kind), [float(v) for v in k]
        self.shape, self.device, self.dtype = tuple(sample.shape), sample.device, sample.dtype

    def node(self) -> "Lin":
        n = Node(self)
        return Lin({id(n): (n, 1.0)}, self.shape, self.device)

    def expanded(self, keep=None):
        return self


def _check_tensor(t: torch.Tensor) -> torch.Tensor:
    if t.device.type not in ("cuda", "cpu"):
        _hip.require_device(t, "sampler operand")
    if t.dtype not in _hip.DTYPE_CODE:
        raise SkrampleHipError(f"unsupported tensor dtype {t.dtype}; the engine handles bf16/f16/f32/f64")
    return t


# ---- host-resident operands ---------------------------------------------------------------------------------------
NUMPY = "numpy"  # `device` of forms built from numpy arrays: evaluated on the host, results handed back as ndarrays
_numpy_views: dict[int, tuple] = {}  # id(ndarray) -> (ndarray, torch view): one leaf per array however often it is lifted


def is_host(device) -> bool:
    return device == NUMPY or (isinstance(device, torch.device) and device.type == "cpu")


def _from_numpy(x) -> torch.Tensor:
    hit = _numpy_views.get(id(x))
    if hit is not None and hit[0] is x:
        return hit[1]
    import numpy as np

    t = torch.from_numpy(np.ascontiguousarray(x))
    if t.dtype not in _hip.DTYPE_CODE:
        raise SkrampleHipError(f"unsupported array dtype {x.dtype}; float16/32/64 arrays are handled")
    if len(_numpy_views) >= 64:
        _numpy_views.pop(next(iter(_numpy_views)))
    _numpy_views[id(x)] = (x, t)
    return t


class Lin:
    "sum_k coef_k * leaf_k ; leaves are HIP tensors, PhiloxNoise or Node objects"

    __slots__ = ("terms", "shape", "device")
    __array_priority__ = 1000

    def __init__(self, terms: dict, shape, device):
        self.terms = terms  # id(leaf) -> (leaf, coef)
        self.shape = tuple(shape)
        self.device = device

    # ---- construction ---------------------------------------------------------------------------
    @staticmethod
    def leaf(obj) -> "Lin":
        if isinstance(obj, torch.Tensor):
            _check_tensor(obj)
            return Lin({id(obj): (obj, 1.0)}, obj.shape, obj.device)
        if isinstance(obj, PhiloxNoise):
            return Lin({id(obj): (obj, 1.0)}, obj.shape, obj.device)
        raise TypeError(type(obj))

    def node(self) -> "Lin":
        "this form, to be stored by the launch that also evaluates its consumers"
        n = Node(self)
        return Lin({id(n): (n, 1.0)}, self.shape, self.device)

    # ---- algebra --------------------------------------------------------------------------------
    def _scaled(self, k: float) -> "Lin":
        return Lin({i: (leaf, c * k) for i, (leaf, c) in self.terms.items()}, self.shape, self.device)

    def _plus(self, other, sign: float) -> "Lin":
        if isinstance(other, NUMBER):
            if other == 0:
                return self
            raise SkrampleHipError("adding a non-zero scalar to a tensor form is not a sampler operation")
        other = lift(other)
        if other.shape != self.shape:
            raise SkrampleHipError(f"shape mismatch in sampler operands: {self.shape} vs {other.shape}")
        terms = dict(self.terms)
        for i, (leaf, c) in other.terms.items():
            if i in terms:
                terms[i] = (leaf, terms[i][1] + sign * c)
            else:
                terms[i] = (leaf, sign * c)
        return Lin(terms, self.shape, self.device)

    def __add__(self, other):
        return self._plus(other, 1.0)

    __radd__ = __add__

    def __sub__(self, other):
        return self._plus(other, -1.0)

    def __rsub__(self, other):
        return self._scaled(-1.0)._plus(other, 1.0)

    def __neg__(self):
        return self._scaled(-1.0)

    def __mul__(self, k):
        if not isinstance(k, NUMBER):
            if hasattr(k, "item") and getattr(k, "ndim", 1) == 0:
                k = float(k.item())
            else:
                raise SkrampleHipError("products of two tensors are not part of any solver step")
        return self._scaled(float(k))

    __rmul__ = __mul__

    def __truediv__(self, k):
        if not isinstance(k, NUMBER):
            raise SkrampleHipError("division by a tensor is not part of any solver step")
        if k == 0:
            raise ZeroDivisionError("tensor form divided by zero")
        return Lin({i: (leaf, c / k) for i, (leaf, c) in self.terms.items()}, self.shape, self.device)

    # ---- expansion ------------------------------------------------------------------------------
    def expanded(self, keep: "Lin | None" = None) -> "Lin":
        "substitute every Node leaf whose form is not `keep`"
        if not any(isinstance(leaf, Node) and leaf.form is not keep for leaf, _ in self.terms.values()):
            return self
        acc = Lin({}, self.shape, self.device)
        for i, (leaf, c) in self.terms.items():
            if isinstance(leaf, Node) and leaf.form is not keep:
                if isinstance(leaf.form, RoundedConversion):
                    raise SkrampleHipError("a rounded conversion can only be consumed by the launch that stores it")
                acc = acc._plus(leaf.form.expanded(keep)._scaled(c), 1.0)
            else:
                acc = acc._plus(Lin({i: (leaf, c)}, self.shape, self.device), 1.0)
        return acc

    def numel(self) -> int:
        return math.prod(self.shape)

    def substitute(self, target, tensor: torch.Tensor) -> "Lin":
        "replace every Node leaf of `target` (a form or RoundedConversion) by the tensor that now holds its value"
        if not any(isinstance(leaf, Node) and leaf.form is target for leaf, _ in self.terms.values()):
            return self
        acc = Lin({}, self.shape, self.device)
        for i, (leaf, c) in self.terms.items():
            piece = Lin.leaf(tensor)._scaled(c) if isinstance(leaf, Node) and leaf.form is target else Lin({i: (leaf, c)}, self.shape, self.device)
            acc = acc._plus(piece, 1.0)
        return acc


def lift(x):
    "number -> number, HIP tensor / PhiloxNoise / Lin -> Lin, anything else -> refused"
    if isinstance(x, Lin):
        return x
    if isinstance(x, NUMBER):
        return x
    if isinstance(x, torch.Tensor):
        if x.ndim == 0 and not x.is_cuda:  # numpy/torch scalars that slipped through float()
            return float(x.item())
        return Lin.leaf(x)
    if isinstance(x, PhiloxNoise):
        return Lin.leaf(x)
    if isinstance(x, LazyTensor):
        return x.form
    if hasattr(x, "dtype") and hasattr(x, "shape") and getattr(x, "shape", None) == ():
        return float(x)
    if type(x).__module__ == "numpy" and hasattr(x, "__array_interface__"):
        t = _from_numpy(x)
        return Lin({id(t): (t, 1.0)}, t.shape, NUMPY)
    raise SkrampleHipError(
        f"operand of type {type(x).__name__} is not supported: skrample_amd computes on torch tensors (HIP device: fused "
        "kernels; CPU: host executor), numpy arrays and host scalars"
    )


# ---------------------------------------------------------------------------------------------------
# evaluation: <= 2 outputs, one kernel launch
# ---------------------------------------------------------------------------------------------------
def _default_dtype(form: Lin) -> torch.dtype:
    for leaf, _ in form.expanded().terms.values():
        if isinstance(leaf, torch.Tensor):
            return leaf.dtype
    return torch.float32


# ---- HBM-aware output placement ---------------------------------------------------------------------------
# A step reads 4-5 equally sized tensors at the same offsets.  torch's caching allocator hands out large blocks
# at 2 MiB multiples, so all streams then start on the same HBM channel/bank phase and collide; shifting the
# tensors this engine allocates (every step result, i.e. the next step's `sample` and history `sample`) by odd
# multiples of 4 KiB takes ~3.5 % off the fused DPM-2 step (tools/tune/tune_step.hip, "mask=21" runs).
_STAGGER_SLOTS = int(os.environ.get("SKR_STAGGER_SLOTS", "8"))  # (environment overrides: tuning experiments only)
_STAGGER_BYTES = int(os.environ.get("SKR_STAGGER_BYTES", "8192"))
_stagger_next = 0


_ITEMSIZE = {torch.bfloat16: 2, torch.float16: 2, torch.float32: 4, torch.float64: 8}
_strides_cache: dict = {}


def empty_output(shape, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    "uninitialised result tensor whose start is shifted by 4 KiB + k*8 KiB (k cycles) inside its allocation"
    global _stagger_next
    shape = tuple(shape)
    numel = math.prod(shape)
    item = _ITEMSIZE.get(dtype, 4)
    if numel * item < (16 << 20):  # small tensors are launch/host-bound and live in L2 / MALL anyway
        return torch.empty(shape, dtype=dtype, device=device)
    k = _stagger_next
    _stagger_next = (k + 1) % _STAGGER_SLOTS
    strides = _strides_cache.get(shape)
    if strides is None:
        acc, rev = 1, []
        for d in reversed(shape):
            rev.append(acc)
            acc *= d
        strides = _strides_cache[shape] = tuple(reversed(rev))
    flat = torch.empty(numel + (4096 + _STAGGER_SLOTS * _STAGGER_BYTES) // item, dtype=dtype, device=device)
    return flat.as_strided(shape, strides, (4096 + k * _STAGGER_BYTES) // item)


def _prepare_tensor(t: torch.Tensor) -> torch.Tensor:
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def evaluate(forms: Sequence[Lin], dtypes: Sequence[torch.dtype | None], acc_f64: bool | None = None) -> list[torch.Tensor]:
    """Materialise one or two forms in a single fused launch.  forms[1] may reference forms[0]
    through a Node leaf (then out1 = chain*out0 + ...)."""
    if not 1 <= len(forms) <= 2:
        raise SkrampleHipError("evaluate() takes one or two forms")
    conv = forms[0] if isinstance(forms[0], RoundedConversion) else None
    if conv is not None:
        if len(forms) != 2:
            raise SkrampleHipError("a rounded conversion is stored alongside the form that consumes it")
        dtypes = [conv.dtype, dtypes[1]]
        # the two operands lead group A; their coef0 is ignored by the kernel
        forms = [conv, forms[1]]
        f0 = Lin({id(conv.sample): (conv.sample, 0.0), id(conv.output): (conv.output, 0.0)}, conv.shape, conv.device)
    else:
        f0 = forms[0].expanded()
    f1 = forms[1].expanded(keep=forms[0]) if len(forms) == 2 else None
    shape, device = f0.shape, f0.device
    numel = math.prod(shape)
    out_dtypes = [d if d is not None else (f.dtype if isinstance(f, RoundedConversion) else _default_dtype(f)) for d, f in zip(dtypes, forms)]
    if is_host(device) or (f1 is not None and is_host(f1.device)):
        return _host_evaluate(conv, f0, f1, out_dtypes, acc_f64)

    # chain coefficient
    chain = 0.0
    if f1 is not None:
        rest = {}
        for i, (leaf, c) in f1.terms.items():
            if isinstance(leaf, Node):
                chain += c
            else:
                rest[i] = (leaf, c)
        f1 = Lin(rest, f1.shape, f1.device)

    # split leaves
    tensors: dict[int, torch.Tensor] = {}
    c0: dict[int, float] = {}
    c1: dict[int, float] = {}
    noise0: list[tuple[PhiloxNoise, float]] = []
    noise1: list[tuple[PhiloxNoise, float]] = []
    for which, form in ((0, f0), (1, f1)):
        if form is None:
            continue
        for i, (leaf, c) in form.terms.items():
            if c == 0.0 and not (conv is not None and which == 0):
                continue
            if isinstance(leaf, PhiloxNoise):
                (noise0 if which == 0 else noise1).append((leaf, c))
            else:
                tensors[i] = leaf
                (c0 if which == 0 else c1)[i] = c

    # in-kernel Philox: one draw per output, fusable shapes only; the rest is realised to tensors
    noise_dtype = torch.float64 if torch.float64 in out_dtypes else torch.float32

    def pick(noises):
        fused, extra = None, []
        for nz, c in noises:
            if fused is None and nz.fusable() and nz.shape == shape:
                fused = (nz, c)
            else:
                extra.append((nz.realize(noise_dtype), c))
        return fused, extra

    fused0, extra0 = pick(noise0)
    fused1, extra1 = pick(noise1)
    if fused0 is not None and fused1 is not None and fused0[0].seeds is not fused1[0].seeds:
        extra1.append((fused1[0].realize(noise_dtype), fused1[1]))
        fused1 = None
    for t, c in extra0:
        tensors[id(t)] = t
        c0[id(t)] = c0.get(id(t), 0.0) + c
    for t, c in extra1:
        tensors[id(t)] = t
        c1[id(t)] = c1.get(id(t), 0.0) + c

    # dtype groups (<= 2); stragglers are cast with a single-term launch of the same kernel
    wide = torch.float64 if any(t.dtype == torch.float64 for t in tensors.values()) or torch.float64 in out_dtypes else torch.float32
    if acc_f64 is None:
        acc_f64 = wide == torch.float64 or _compute_dtype.get() == torch.float64
    if acc_f64:
        wide = torch.float64
    prepared = {i: _prepare_tensor(t) for i, t in tensors.items()}
    for t in prepared.values():
        if t.shape != shape and t.numel() != numel:
            raise SkrampleHipError(f"operand shape {tuple(t.shape)} does not match {shape}")
    narrow_types = [t.dtype for t in prepared.values() if t.dtype != wide]
    group_a = max(dict.fromkeys(narrow_types), key=narrow_types.count) if narrow_types else wide  # (first seen wins a tie: no dependence on hash order)
    if conv is not None and conv.dtype == wide:
        group_a = wide  # the rounded conversion's operands lead group A: whatever is narrower (fp32 noise beside fp64 latents) is widened, exactly, below
    # outputs must be group_a or wide
    for k, od in enumerate(out_dtypes):
        if od not in (group_a, wide):
            if not narrow_types or all(d == wide for d in narrow_types):
                group_a = od  # no narrow inputs: the narrow slot is free for the output dtype
            else:
                raise SkrampleHipError(f"output dtype {od} incompatible with operand dtypes {group_a}/{wide}")
    for i, t in list(prepared.items()):
        if t.dtype not in (group_a, wide):
            prepared[i] = cast(t, wide)
    ids_a = [i for i, t in prepared.items() if t.dtype == group_a]
    ids_b = [i for i, t in prepared.items() if t.dtype != group_a]
    if conv is not None:
        if group_a != conv.dtype:
            raise SkrampleHipError("rounded conversion operands must form the narrow dtype group")
        lead = [id(conv.sample), id(conv.output)]
        ids_a = lead + [i for i in ids_a if i not in lead]
    order = ids_a + ids_b
    if len(order) > _hip.MAX_TERMS:
        raise SkrampleHipError(f"{len(order)} operands exceed the kernel limit of {_hip.MAX_TERMS}")

    plan = _hip.StepPlanC()
    plan.n_terms = len(order)
    plan.n_group_a = len(ids_a)
    plan.dtype_a = _hip.DTYPE_CODE[group_a]
    plan.dtype_b = _hip.DTYPE_CODE[wide]
    plan.out0_dtype = _hip.DTYPE_CODE[out_dtypes[0]]
    plan.out1_dtype = _hip.DTYPE_CODE[out_dtypes[1]] if f1 is not None else _hip.NONE
    plan.acc_f64 = 1 if acc_f64 else 0
    for k, i in enumerate(order):
        plan.coef0[k] = c0.get(i, 0.0)
        plan.coef1[k] = c1.get(i, 0.0)
    plan.chain = chain
    if conv is not None:
        plan.convert_to, plan.convert_from = conv.to_kind, conv.from_kind
        for k in range(4):
            plan.convert_k[k] = conv.k[k]
    seeds = None
    if fused0 is not None or fused1 is not None:
        plan.noise_mode = 1
        src = fused0 or fused1
        seeds = src[0].seeds
        plan.sample_numel = src[0].sample_numel
        if fused0 is not None:
            plan.zeta0, plan.stream0 = fused0[1], fused0[0].stream
        if fused1 is not None:
            plan.zeta1, plan.stream1 = fused1[1], fused1[0].stream
    out0 = empty_output(shape, out_dtypes[0], device)
    out1 = empty_output(shape, out_dtypes[1], device) if f1 is not None else None
    _hip.launch_step(plan, [prepared[i] for i in order], out0, out1, seeds, numel, device)
    return [out0] if out1 is None else [out0, out1]



def _host_evaluate(conv, f0: "Lin", f1, out_dtypes, acc_f64) -> list:
    """Host executor for host-resident operands (CPU torch tensors / numpy arrays): the same one or two forms, evaluated
    with plain torch ops in the accumulator precision the kernel uses (fp32, or fp64 when asked for), each output rounded
    once.  Not a fallback: forms over HIP tensors never come here, and a form that mixes residencies is refused."""
    as_numpy = f0.device == NUMPY or (f1 is not None and f1.device == NUMPY)
    leaves = [leaf for form in (f0, f1) if form is not None for leaf, _ in form.terms.values() if isinstance(leaf, torch.Tensor)]
    if conv is not None:
        leaves += [conv.sample, conv.output]
    for t in leaves:
        if t.device.type != "cpu":
            raise SkrampleHipError("operands of one step must all live on the HIP device or all on the host")
    wide = torch.float64 if acc_f64 or _compute_dtype.get() == torch.float64 or torch.float64 in out_dtypes or any(t.dtype == torch.float64 for t in leaves) else torch.float32

    def total(form, start=None):
        acc = start
        for leaf, c in form.terms.values():
            if isinstance(leaf, Node):
                continue  # the chained term is added by the caller
            if isinstance(leaf, PhiloxNoise):
                raise SkrampleHipError("in-kernel Philox noise exists on the HIP device only")
            if c == 0.0:
                continue
            term = leaf.to(wide) * c
            acc = term if acc is None else acc.add_(term)
        return acc if acc is not None else torch.zeros(form.shape, dtype=wide)

    if conv is not None:
        # the reference's op-by-op conversion in the operands' own dtype (diffusers.py:819-834, models.py:92-224)
        s_, o_, k = conv.sample, conv.output, conv.k
        x = {0: lambda: o_, 1: lambda: (s_ - k[0] * o_) / k[1], 2: lambda: k[1] * s_ - k[0] * o_, 3: lambda: o_ * k[0]}[conv.to_kind]()
        v0 = {0: lambda: x, 1: lambda: (s_ - k[2] * x) / k[3], 2: lambda: (k[2] * s_ - x) / k[3], 3: lambda: x / k[2]}[conv.from_kind]()
        acc0 = v0.to(wide)
    else:
        acc0 = total(f0)
    outs = [acc0.to(out_dtypes[0])]
    if f1 is not None:
        chain = sum(c for leaf, c in f1.terms.values() if isinstance(leaf, Node))
        acc1 = total(f1, acc0 * chain if chain != 0.0 else None)
        outs.append(acc1.to(out_dtypes[1]))
    return [o.numpy() for o in outs] if as_numpy else outs


def cast(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    "dtype conversion through the engine (a one-term fused launch)"
    if t.dtype == dtype:
        return t
    return evaluate([Lin.leaf(t)], [dtype])[0]


_norm_ws: dict = {}


def error_mean(a, b: torch.Tensor, power: int) -> float:
    "mean(|a - b|^power) over a device tensor pair (a may be the number 0); one reduction launch + read-back (ndarrays: on the host)"
    a = _from_numpy(a) if type(a).__module__ == "numpy" and hasattr(a, "__array_interface__") and getattr(a, "ndim", 0) > 0 else a
    b = _from_numpy(b) if type(b).__module__ == "numpy" and hasattr(b, "__array_interface__") else b
    _check_tensor(b)
    if not b.is_cuda:  # host-resident operands: plain torch (fp64 accumulation, as the kernel)
        d = (b.double() if not isinstance(a, torch.Tensor) else a.double() - b.double()).abs()
        return float((d if power == 1 else d.pow(power)).mean())
    if isinstance(a, torch.Tensor):
        _check_tensor(a)
        if a.dtype != b.dtype or a.shape != b.shape:
            raise SkrampleHipError("error norm operands must share dtype and shape")
        a = _prepare_tensor(a)
    elif a != 0:
        raise SkrampleHipError("error norm against a non-zero scalar is not supported")
    b = _prepare_tensor(b)
    ws = _norm_ws.get(b.device)
    if ws is None:
        ws = _norm_ws[b.device] = torch.empty(1025, dtype=torch.float64, device=b.device)
    lib = _hip.load()
    status = lib.skr_error_mean(a.data_ptr() if isinstance(a, torch.Tensor) else None, b.data_ptr(), _hip.DTYPE_CODE[b.dtype], b.numel(), power, ws.data_ptr(), ws.data_ptr() + 8, _hip.current_stream_ptr(b.device))
    _hip.check(status, "skr_error_mean")
    return ws[0].item()


def power_blend(a: torch.Tensor, b: torch.Tensor, wa: float, wb: float, power: float, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    "spowf(wa * spowf(a, power) + wb * spowf(b, power), 1 / power) as one elementwise launch; fp32 or fp64 result (ndarrays in: ndarray out)"
    as_numpy = not isinstance(a, torch.Tensor) or not isinstance(b, torch.Tensor)
    a = a if isinstance(a, torch.Tensor) else _from_numpy(a)
    b = b if isinstance(b, torch.Tensor) else _from_numpy(b)
    _check_tensor(a)
    _check_tensor(b)
    if a.shape != b.shape:
        raise SkrampleHipError(f"shape mismatch in sampler operands: {tuple(a.shape)} vs {tuple(b.shape)}")
    if dtype not in (torch.float32, torch.float64):
        raise SkrampleHipError("the signed-power blend is evaluated in float32 or float64")
    if not a.is_cuda and not b.is_cuda:  # host-resident operands
        spow = lambda v, f: v.abs().pow(f) * v.sign()  # noqa: E731
        blended = spow(wa * spow(a.to(dtype), power) + wb * spow(b.to(dtype), power), 1 / power)
        return blended.numpy() if as_numpy else blended
    a, b = _prepare_tensor(a), _prepare_tensor(b)
    out = empty_output(a.shape, dtype, a.device)
    status = _hip.load().skr_power_blend(out.data_ptr(), _hip.DTYPE_CODE[dtype], a.data_ptr(), _hip.DTYPE_CODE[a.dtype], b.data_ptr(), _hip.DTYPE_CODE[b.dtype], float(wa), float(wb), float(power), a.numel(), _hip.current_stream_ptr(a.device))
    _hip.check(status, "skr_power_blend")
    return out


def settle(value, like=None, dtype: torch.dtype | None = None):
    "number -> number; form -> tensor (one launch)"
    if isinstance(value, Lin):
        if dtype is None and isinstance(like, torch.Tensor):
            dtype = like.dtype
        return evaluate([value], [dtype])[0]
    return value


class LazyTensor:
    """A form that is evaluated on first use.  Returned where the reference returns a tensor that
    callers rarely read (e.g. `pred_original_sample`): unused, it costs no HBM traffic; used in any
    torch function, arithmetic, indexing or via `.materialize()`, it becomes an ordinary tensor.

    The form's leaves are the caller's own tensors, so a late evaluation would silently use whatever they hold by
    then: every leaf is stamped (data_ptr, _version) at creation and `materialize()` refuses to run once one of them
    has been modified in place -- read the value before reusing the buffers, as with any view."""

    def __init__(self, form: Lin | None, dtype: torch.dtype, form_fn=None, shape=None, device=None, leaves=None, acc_f64: bool | None = None):
        self._form, self._form_fn, self.dtype, self._value = form, form_fn, dtype, None
        # the accumulator of the step that made this tensor (compute_scale=float64): it is read later, outside that step's context
        self._acc_f64 = (_compute_dtype.get() == torch.float64) if acc_f64 is None else bool(acc_f64)
        self._shape, self._device = shape, device
        if leaves is None and form is not None and isinstance(form, Lin):
            leaves = [leaf for leaf, _ in form.terms.values() if isinstance(leaf, torch.Tensor)]
        self._stamps = [(t, t.data_ptr(), t._version) for t in (leaves or ()) if isinstance(t, torch.Tensor)]

    @property
    def form(self) -> Lin:
        if self._form is None:
            self._form = self._form_fn()
        return self._form

    @property
    def shape(self):
        return self._shape if self._form is None and self._shape is not None else self.form.shape

    @property
    def device(self):
        return self._device if self._form is None and self._device is not None else self.form.device

    def materialize(self) -> torch.Tensor:
        if self._value is None:
            for t, ptr, version in self._stamps:
                if t._version != version or t.data_ptr() != ptr:
                    raise SkrampleHipError("an operand of this lazily evaluated tensor was modified in place before it was read; materialize() it (or use it) before reusing the buffers it was computed from")
            self._value = evaluate([self.form], [self.dtype], acc_f64=True if self._acc_f64 else None)[0]
            self._stamps = []
        return self._value

    def to(self, *args, **kwargs):
        return self.materialize().to(*args, **kwargs)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    # python operators and the container protocol are looked up on the type, not through __getattr__
    def __len__(self):
        return len(self.materialize())

    def __getitem__(self, key):
        return self.materialize()[key]

    def __iter__(self):
        return iter(self.materialize())

    def __neg__(self):
        return -self.materialize()

    def __abs__(self):
        return abs(self.materialize())

    def __array__(self, dtype=None):
        v = self.materialize().detach().cpu()
        return v.float().numpy() if v.dtype in (torch.bfloat16,) else (v.numpy() if dtype is None else v.numpy().astype(dtype))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def unwrap(a):
            if isinstance(a, LazyTensor):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(unwrap(v) for v in a)
            return a

        return func(*unwrap(args), **{k: unwrap(v) for k, v in (kwargs or {}).items()})


def _lazy_binary(name: str):
    def op(self, other):
        other = other.materialize() if isinstance(other, LazyTensor) else other
        return getattr(self.materialize(), name)(other)

    op.__name__ = name
    return op


for _name in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "pow", "rpow", "matmul", "rmatmul", "floordiv", "mod", "eq", "ne", "lt", "le", "gt", "ge"):
    setattr(LazyTensor, f"__{_name}__", _lazy_binary(f"__{_name}__"))
LazyTensor.__hash__ = object.__hash__  # (defining __eq__ would otherwise drop hashability)


def is_number(x: Any) -> bool:
    return isinstance(x, NUMBER)
