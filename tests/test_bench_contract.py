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


def test_round4_lines_of_every_config_follow_the_contract():
    "one committed line per BASELINE config (profiles/r04_bench_line*.json, `python bench.py --config <c>` on an MI355X)"
    expect = {"": (256, 10), "_k20": (256, 10), "_cfg2": (64, 10), "_cfg3": (256, 26), "_cfg3c": (256, 30), "_cfg4": (256, 18), "_cfg5": (64, 100)}
    units = {"_cfg3": 16 * 128 * 128, "_cfg3c": 16 * 128 * 128, "_cfg5": 4 * 256 * 256}
    for tag, (batch, bytes_per_elem) in expect.items():
        d = line(f"r04_bench_line{tag}.json")
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in d, (tag, key)
        assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["unit"] == "steps/s"
        assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]  # whole-job steps / wall time
        assert d["config"]["per_gpu_batch"] == batch and "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        numel = batch * units.get(tag, 4 * 128 * 128)
        assert r["algorithmic_bytes_per_step"] == numel * bytes_per_elem and r["algorithmic_bytes_per_element"] == bytes_per_elem
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (r["us_per_step"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
        assert r["traffic"] is not None and r["traffic_source"].startswith("live")
        if tag != "_cfg5":  # cfg5 moves FEWER bytes than 8(d) counts (derivatives stored as 2 B, counted as 4 B pairs): stated in DESIGN
            assert 0.98 < r["traffic"] / r["algorithmic_bytes_per_step"] < 1.05
        else:
            assert 0.80 < r["traffic"] / r["algorithmic_bytes_per_step"] < 1.0
        assert ("whole_step" in r) == (tag in ("_cfg3c", "_cfg5"))  # configs that name a noise generator report it separately
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "steps/s" and c["sample"]


def test_round5_lines_of_every_config_follow_the_contract():
    "profiles/r05_bench_line*.json (one gpurun call, `tools/collect_r05.sh bench`): the contract keys, the roofline arithmetic, per-rank clocks"
    expect = {"": (256, 10), "_k20": (256, 10), "_cfg2": (64, 10), "_cfg3": (256, 26), "_cfg3c": (256, 30), "_cfg4": (256, 18), "_cfg5": (64, 100)}
    units = {"_cfg3": 16 * 128 * 128, "_cfg3c": 16 * 128 * 128, "_cfg5": 4 * 256 * 256}
    for tag, (batch, bytes_per_elem) in expect.items():
        d = line(f"r05_bench_line{tag}.json")
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in d, (tag, key)
        assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["unit"] == "steps/s"
        assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
        r = d["roofline"]
        numel = batch * units.get(tag, 4 * 128 * 128)
        assert r["algorithmic_bytes_per_step"] == numel * bytes_per_elem and r["bound"] == "hbm" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (r["us_per_step"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
        assert r["traffic"] is not None and r["traffic_source"].startswith("live") and abs(r["frac_on_measured_traffic"] - r["traffic"] / (r["us_per_step"] * 1e-6) / 1e9 / 8000.0) < 1e-9
        ranks = r["ranks"]  # every rank's own clocks, read before the closing barrier (one rank here)
        assert ranks["n_ranks_seen"] == 1 and len(ranks["wall_us_per_step"]["per_rank"]) == 1
        assert abs(ranks["wall_us_per_step"]["max"] - d["ms_per_step"] * 1e3) < 1e-6 * ranks["wall_us_per_step"]["max"]
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # the two-rank rehearsal at the driver's K = 20 / W = 5: both ranks seen, value from the slower rank's own wall clock
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r05_two_rank_rehearsal.txt")) if l.startswith("{")]
    assert len(rows) == 3
    single = rows.pop()  # the RCCL path itself (backend nccl) with one rank under torch.distributed.run
    assert single["backend"].startswith("nccl") and single["n_gpus"] == 1 and single["roofline.ranks"]["n_ranks_seen"] == 1 and single["steps"] == 20
    for d in rows:
        ranks = d["roofline.ranks"]
        assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and ranks["n_ranks_seen"] == 2 and len(ranks["wall_us_per_step"]["per_rank"]) == 2
        assert abs(d["ms_per_step"] * 1e3 - ranks["wall_us_per_step"]["max"]) < 1e-6 * ranks["wall_us_per_step"]["max"]
        assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]  # whole-job rate: both ranks' steps over the slower rank's time


def test_bench_cli_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config"):
        assert flag in out.stdout
