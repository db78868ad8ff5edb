"""The bench line's contract (task statement, "Measurement"): one JSON object with the metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config keys plus `roofline` and `cpu_baseline`.  Checked on the
committed line of the round (profiles/r02_bench_line.json, produced on an MI355X by `python bench.py`) and on bench.py's own CLI."""

import json
import os
import subprocess
import sys

from conftest import ROOT


def line(name):
    return json.loads(open(os.path.join(ROOT, "profiles", name)).read())


def test_committed_bench_lines_follow_the_contract():
    for name, steps in (("r02_bench_line.json", 400), ("r02_bench_line_k20.json", 20)):
        d = line(name)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in d, (name, key)
        assert d["steps"] == steps and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
        assert d["unit"] == "steps/s" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
        assert abs(d["value"] - d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]  # value = whole-job steps / wall time
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_us_per_launch"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
        assert r["algorithmic_bytes_per_launch"] == 256 * 4 * 128 * 128 * 10  # 10 B per element
        assert r["traffic"] is None or 0.98 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05  # PMC bytes ~ algorithmic bytes
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "steps/s" and c["sample"]
        assert d["config"]["per_gpu_batch"] == 256 and d["config"]["global_batch"] == 256


def test_bench_cli_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
