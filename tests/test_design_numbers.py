"""DESIGN.md / README.md quote measured numbers; the tables of measured numbers are GENERATED from the committed evidence files
(tools/design_tables.py) so that a quoted figure cannot drift from the file it cites.  This test re-generates them and fails when
the documents differ, and checks that every file a generated row names exists and carries that number."""

import csv
import json
import os
import re
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import design_tables  # noqa: E402


def test_generated_tables_match_the_committed_profiles():
    for fname, names in design_tables.TARGETS.items():
        text = open(os.path.join(ROOT, fname)).read()
        assert design_tables.render(text, names) == text, f"{fname}: run `python tools/design_tables.py` after changing profiles/"


def test_every_bench_row_names_a_committed_line_with_that_value():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    block = re.search(r"<!-- GENERATED:bench_lines BEGIN[^>]*-->\n(.*?)\n<!-- GENERATED:bench_lines END -->", text, re.S).group(1)
    rows = [r for r in block.splitlines()[2:] if r.startswith("|")]
    assert len(rows) >= 7, "headline (K=400 and K=20) + cfg2 / cfg3 / cfg3c / cfg4 / cfg5"
    for row in rows:
        cells = [c.strip() for c in row.strip("|").split("|")]
        path = re.search(r"`(profiles/[^`]+)`", cells[-1]).group(1)
        d = json.loads(open(os.path.join(ROOT, path)).read().strip().splitlines()[-1])
        assert f"{d['value']:.1f}" == cells[1], (path, cells[1])
        assert f"{d['roofline']['frac']:.3f}" == cells[4], (path, cells[4])
        assert cells[5] == ("—" if d["roofline"].get("frac_on_measured_traffic") is None else f"{d['roofline']['frac_on_measured_traffic']:.3f}"), (path, cells[5])
        # the contract keys of every line
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in d, (path, key)
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (r["us_per_step"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
        assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]


def test_every_kernel_row_is_a_row_of_its_csv():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    block = re.search(r"<!-- GENERATED:kernel_rows BEGIN[^>]*-->\n(.*?)\n<!-- GENERATED:kernel_rows END -->", text, re.S).group(1)
    rows = [r for r in block.splitlines()[2:] if r.startswith("|")]
    assert rows, "no kernel rows: profiles/<round>_kernel_stats_<config>.csv missing?"
    seen = set()
    for row in rows:
        cells = [c.strip() for c in row.strip("|").split("|")]
        cfg, kernel, avg = cells[0], cells[1].strip("`"), cells[4]
        seen.add(cfg)
        csv_rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{design_tables.ROUND}_kernel_stats_{cfg}.csv"))))
        hits = [r for r in csv_rows if design_tables.short_kernel(r["Name"]) == kernel and f"{float(r['AverageNs']) / 1e3:.2f}" == avg]
        assert hits, (cfg, kernel, avg)
    assert seen == set(design_tables.CONFIGS)
