"""The drop-in boundary on a box without a GPU: the C-ABI library loads and exports every symbol
the header declares, the product never imports the oracle, and nothing falls back to the CPU."""

import ast
import ctypes
import os
import re

import pytest
import torch
from conftest import ROOT

from skrample_amd import _hip


def test_library_exports_header_symbols():
    header = open(os.path.join(ROOT, "include", "skrample_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|void|const char\*)\s+(skr_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_hip.EXPORTS), (declared, set(_hip.EXPORTS))
    assert os.path.isfile(_hip.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.skr_abi_version.restype = ctypes.c_int
    assert lib.skr_abi_version() == _hip.ABI_VERSION == int(re.search(r"#define SKR_ABI_VERSION (\d+)", header).group(1))
    lib.skr_strerror.restype = ctypes.c_char_p
    assert lib.skr_strerror(0) == b"ok" and b"aligned" in lib.skr_strerror(4)
    assert int(re.search(r"#define SKR_MAX_TERMS (\d+)", header).group(1)) == _hip.MAX_TERMS


def test_plan_struct_layout_matches_header():
    "ctypes mirror of skr_step_plan: 8 int32, 2*MAX doubles, 3 doubles, 2 uint64, 1 int64, 2 int32, 4 doubles"
    assert ctypes.sizeof(_hip.StepPlanC) == 8 * 4 + 2 * _hip.MAX_TERMS * 8 + 3 * 8 + 2 * 8 + 8 + 2 * 4 + 4 * 8
    assert _hip.StepPlanC.coef0.offset == 32 and _hip.StepPlanC.convert_k.offset == ctypes.sizeof(_hip.StepPlanC) - 32


def test_argument_validation_without_gpu():
    "host-side argument checks run before any launch"
    lib = _hip.load()
    plan = _hip.StepPlanC()
    assert lib.skr_step_launch(None, None, None, None, None, 8, None) == 1  # SKR_ERR_NULL
    plan.n_terms = _hip.MAX_TERMS + 1
    assert lib.skr_step_launch(ctypes.byref(plan), None, None, None, None, 8, None) == 3  # SKR_ERR_TERMS
    plan.n_terms, plan.n_group_a, plan.out0_dtype, plan.out1_dtype = 0, 0, _hip.F32, _hip.NONE
    assert lib.skr_step_launch(ctypes.byref(plan), None, None, None, None, -1, None) == 5  # SKR_ERR_SHAPE
    assert lib.skr_step_launch(ctypes.byref(plan), None, None, None, None, 0, None) == 0  # empty batch is a no-op
    assert lib.skr_step_launch(ctypes.byref(plan), None, None, None, None, 8, None) == 1  # out0 missing
    buf = (ctypes.c_char * 64)()
    addr = ctypes.addressof(buf)
    misaligned = addr + (8 if addr % 16 == 0 else 16 - addr % 16 + 8)
    assert lib.skr_step_launch(ctypes.byref(plan), None, ctypes.c_void_p(misaligned), None, None, 8, None) == 4  # SKR_ERR_ALIGN
    assert lib.skr_noise_random(None, _hip.F32, None, 0, 0, 16, None) == 0
    assert lib.skr_noise_random(None, _hip.F32, None, 0, 2, 16, None) == 1
    assert lib.skr_philox_u32(None, 0, 0, 0, 4, None) == 1


def _imports(path: str) -> set[str]:
    tree = ast.parse(open(path).read())
    names = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            names |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
            names.add(node.module.split(".")[0])
    return names


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "skrample_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                path = os.path.join(dirpath, f)
                assert not ({"skr_oracle", "oracle"} & _imports(path)), path
    # nor do the tools (benchmarks, profilers, fixture generators): the checker scripts that call the oracle live in tests/
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            text = open(os.path.join(ROOT, "tools", f)).read()
            assert "skr_oracle" not in text and '"oracle"' not in text and "'oracle'" not in text, f


def test_host_operands_stay_on_the_host_and_never_need_the_library(monkeypatch, tmp_path):
    """CPU tensors are the reference's host path (generic T, BASELINE config 1): they are computed by the package's own
    host executor without touching the HIP library -- which may even be absent -- and nothing about them is a fallback for
    device work (tests/test_host_path.py::test_device_work_never_reaches_the_host_executor)."""
    import skrample_amd.scheduling as PS
    from skrample_amd.sampling import lazy, models, structured

    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "libskrample_hip.so"))  # no library: host operands must not care
    x = torch.randn(1, 4, 8, 8)
    rec = structured.Euler().sample(x, x, (0.0, 0.1), models.NoiseModel(), PS.Scaled())
    assert isinstance(rec.final, torch.Tensor) and rec.final.device.type == "cpu" and torch.isfinite(rec.final).all()
    assert lazy.cast(x, torch.bfloat16).dtype == torch.bfloat16
    with pytest.raises(_hip.SkrampleHipError):  # unsupported dtypes are still refused
        lazy.cast(torch.ones(4, dtype=torch.int32), torch.float32)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "libskrample_hip.so"))
    with pytest.raises(_hip.SkrampleHipError, match="no CPU fallback"):
        _hip.load()


def test_header_is_plain_c(tmp_path):
    "the drop-in boundary is a C ABI: include/skrample_hip.h must compile as C99 (and as C++) with nothing but the standard headers"
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    src = tmp_path / "h.c"
    src.write_text(f'#include "{os.path.join(ROOT, "include", "skrample_hip.h")}"\nint main(void) {{ skr_step_plan p; (void)p; return SKR_ABI_VERSION == {_hip.ABI_VERSION} ? 0 : 1; }}\n')
    assert subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-o", str(tmp_path / "h.o")], capture_output=True).returncode == 0
    if shutil.which("g++") is not None:
        assert subprocess.run(["g++", "-std=c++17", "-Wall", "-fsyntax-only", "-x", "c++", str(src)], capture_output=True).returncode == 0


@pytest.mark.gpu
def test_plain_c_client(tmp_path):
    """tests/abi_client.c -- a C99 program that knows only include/skrample_hip.h, the HIP runtime and libskrample_hip.so (no Python,
    no torch) -- builds with gcc and runs a one-output step, a two-output step and an in-kernel Philox draw through the ABI"""
    import shutil
    import subprocess

    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the ROCm headers")
    exe = tmp_path / "abi_client"
    lib_dir = os.path.dirname(_hip.LIB_PATH)
    build = subprocess.run(
        ["gcc", "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_client.c"),
         "-L", lib_dir, "-lskrample_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
        capture_output=True, text=True)  # fmt: skip
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "abi client ok" in run.stdout, (run.stdout, run.stderr)
