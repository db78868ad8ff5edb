#!/usr/bin/env python3
"""Generate the measured-number tables of DESIGN.md / README.md from the COMMITTED evidence files, so that no quoted figure can
drift from the file it cites ("DESIGN quotes numbers the committed profile does not contain").

  inputs : profiles/r05_bench_line[_k20|_<config>].json   (one JSON line per `python bench.py [--config <c>]`)
           profiles/r05_kernel_stats_<config>.csv          (tools/kernel_stats.py over `rocprofv3 --kernel-trace --stats -- python3 bench.py --config <c> ...`)
  output : the text between  <!-- GENERATED:<name> BEGIN ... -->  and  <!-- GENERATED:<name> END -->  in DESIGN.md and README.md

usage:  python tools/design_tables.py            rewrite the blocks in place
        python tools/design_tables.py --check    exit 1 if a block differs from what the committed files give
tests/test_design_numbers.py runs the check in the CPU suite.
"""
from __future__ import annotations

import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROFILES = os.path.join(ROOT, "profiles")
ROUND = "r05"
CONFIGS = ("headline", "cfg2", "cfg3", "cfg3c", "cfg4", "cfg5")


def bench_line(name: str) -> dict | None:
    path = os.path.join(PROFILES, name)
    if not os.path.isfile(path):
        return None
    text = open(path).read().strip()
    return json.loads(text.splitlines()[-1]) if text else None


def short_kernel(name: str) -> str:
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)  # parameter list
    return name.replace("skr::", "")


def template_ints(name: str) -> list[int]:
    inner = name[name.index("<") + 1 :] if "<" in name else ""
    return [int(v) for v in re.findall(r"(?<![\w.])(\d+)(?![\w.])", inner)]


def step_kernel_bytes_per_element(short: str) -> int | None:
    "algorithmic bytes per element of ONE launch of a bf16 step kernel, from its template arguments (SURVEY 8(d) counting)"
    ints = template_ints(short)
    if short.startswith("step_kernel_k1<") and ints:
        return 2 * ints[0] + 2  # K bf16 operands + bf16 result
    if short.startswith("step_kernel_k2<") and len(ints) >= 2:
        return 2 * ints[0] + 4 * ints[1] + 4 + 2  # NA bf16 + NB fp32 operands, fp32 state + bf16 result
    if short.startswith("step_kernel_rk1<") and ints:
        return 2 * ints[0] + 4  # K bf16 operands, derivative + next stage input (bf16 each)
    return None


def kernel_rows(config: str) -> list[dict]:
    path = os.path.join(PROFILES, f"{ROUND}_kernel_stats_{config}.csv")
    if not os.path.isfile(path):
        return []
    return list(csv.DictReader(open(path)))


def fmt(v: float, digits: int = 1) -> str:
    return f"{v:.{digits}f}"


def table_lines() -> str:
    "one row per committed bench line"
    rows = [
        "| config (`bench.py --config`) | `value` steps/s | µs per step (wall) | step kernels µs (HIP events) | `roofline.frac` (8(d) bytes ÷ 8 TB/s) | `frac_on_measured_traffic` (PMC bytes ÷ 8 TB/s) | whole step incl. generator (8(d) bytes ÷ 8 TB/s) | by wall clock | PMC traffic ÷ algorithmic | eager wrapper steps/s | CPU baseline steps/s (threads) | file |",
        "|---|---|---|---|---|---|---|---|---|---|---|---|",
    ]
    files = [("headline", f"{ROUND}_bench_line.json"), ("headline, driver's K=20 / W=5", f"{ROUND}_bench_line_k20.json")] + [(c, f"{ROUND}_bench_line_{c}.json") for c in CONFIGS[1:]]
    for label, name in files:
        d = bench_line(name)
        if d is None:
            continue
        r = d["roofline"]
        ratio = "—" if not r.get("traffic") else fmt(r["traffic"] / r["algorithmic_bytes_per_step"], 4)
        cpu = d.get("cpu_baseline") or {}
        wrap = d.get("wrapper_steps_per_s")
        whole = r.get("whole_step")
        extra = f" (whole step with generator: {fmt(whole['us_per_step'])} µs, generator {fmt(whole['generator_us_per_step'])} µs)" if whole else ""
        measured = r.get("frac_on_measured_traffic")
        rows.append(
            f"| {label} | {fmt(d['value'])} | {fmt(d['ms_per_step'] * 1e3, 2)} | {fmt(r['us_per_step'], 2)}{extra} | {fmt(r['frac'], 3)} | {'—' if measured is None else fmt(measured, 3)} | "
            f"{'—' if not whole else fmt(whole['frac_of_step_bytes'], 3)} | {fmt(r['wall_clock']['frac'], 3)} | {ratio} | "
            f"{'—' if wrap is None else fmt(wrap)} | {'—' if not cpu else fmt(cpu['value'], 2) + ' (' + str(cpu['cores']) + ')'} | `profiles/{name}` |"
        )
    return "\n".join(rows)


def table_kernels() -> str:
    "one row per (config, kernel) of the committed rocprofv3 kernel-trace summaries"
    import bench  # the workload table (names, shapes, kernel needles); imports torch, no GPU needed

    workloads = bench._wl()
    rows = [
        "| config | kernel (row of the CSV) | workgroups × threads | calls | avg µs | min – max µs | algorithmic bytes per launch → rate at the average (fraction of 8 TB/s) |",
        "|---|---|---|---|---|---|---|",
    ]
    for c in CONFIGS:
        wl = workloads[c]
        numel = wl.batch * wl.unit[0] * wl.unit[1] * wl.unit[2]
        needles = [n for n, _ in wl.pmc_kernels]
        for r in kernel_rows(c):
            if not any(n in r["Name"] for n in needles) or int(r["Calls"]) < 10:
                continue
            short = short_kernel(r["Name"])
            avg_us = float(r["AverageNs"]) / 1e3
            per = step_kernel_bytes_per_element(short)
            if per is not None and int(r["Workgroups"]) * int(r["WorkgroupSize"]) * 8 == numel:
                gbs = per * numel / (avg_us * 1e-6) / 1e9
                rate = f"{per} B × {numel / 1e6:.1f} M = {per * numel / 1e6:.1f} MB → {gbs / 1e3:.2f} TB/s ({gbs / 8000:.3f})"
            else:
                rate = "—"
            rows.append(f"| {c} | `{short}` | {r['Workgroups']} × {r['WorkgroupSize']} | {r['Calls']} | {fmt(avg_us, 2)} | {fmt(int(r['MinNs']) / 1e3, 2)} – {fmt(int(r['MaxNs']) / 1e3, 2)} | {rate} |")
    return "\n".join(rows)


def table_noise() -> str:
    "generator kernels of tools/prof_noise.py (256 x (16,128,128) bf16 draws of every generator), rows of profiles/r05_noise_kernel_stats.csv"
    path = os.path.join(PROFILES, f"{ROUND}_noise_kernel_stats.csv")
    if not os.path.isfile(path):
        return "(no committed noise kernel stats yet)"
    rows = [f"| kernel (row of `profiles/{ROUND}_noise_kernel_stats.csv`) | workgroups × threads | calls | avg µs | min – max µs |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(path)):
        short = short_kernel(r["Name"])
        if short.startswith(("at::", "__amd", "void at::")) or "at::native" in r["Name"]:
            continue
        rows.append(f"| `{short[:110]}` | {r['Workgroups']} × {r['WorkgroupSize']} | {r['Calls']} | {fmt(float(r['AverageNs']) / 1e3, 2)} | {fmt(int(r['MinNs']) / 1e3, 2)} – {fmt(int(r['MaxNs']) / 1e3, 2)} |")
    return "\n".join(rows)


def table_readme() -> str:
    d, k = bench_line(f"{ROUND}_bench_line.json"), bench_line(f"{ROUND}_bench_line_k20.json")
    if d is None:
        return "(no committed bench line yet)"
    r = d["roofline"]
    out = [
        f"* headline (`python bench.py`, {d['steps']} steps): **{fmt(d['value'])} steps/s**, {fmt(r['us_per_step'], 2)} µs per launch = {fmt(r['achieved'] / 1e3, 2)} TB/s algorithmic = "
        f"**{fmt(100 * r['frac'], 1)} %** of the 8 TB/s spec ({fmt(100 * r['wall_clock']['frac'], 1)} % by the wall clock `value` is computed from); PMC traffic ÷ algorithmic bytes = {fmt(r['traffic'] / r['algorithmic_bytes_per_step'], 4)}"
    ]
    if k is not None:
        out.append(f"* the driver's K=20 / W=5 run of the same command: {fmt(k['value'])} steps/s, {fmt(100 * k['roofline']['frac'], 1)} % by the event clock, {fmt(100 * k['roofline']['wall_clock']['frac'], 1)} % by wall clock")
    for c in CONFIGS[1:]:
        e = bench_line(f"{ROUND}_bench_line_{c}.json")
        if e is not None:
            out.append(f"* `--config {c}`: {fmt(e['value'])} steps/s, step kernels {fmt(e['roofline']['us_per_step'], 1)} µs = {fmt(100 * e['roofline']['frac'], 1)} % of 8 TB/s on the algorithmic bytes")
    return "\n".join(out)


BLOCKS = {"bench_lines": table_lines, "kernel_rows": table_kernels, "noise_rows": table_noise, "readme": table_readme}
TARGETS = {"DESIGN.md": ("bench_lines", "kernel_rows", "noise_rows"), "README.md": ("readme",)}


def render(text: str, names) -> str:
    for name in names:
        pattern = re.compile(rf"(<!-- GENERATED:{name} BEGIN[^>]*-->\n).*?(\n<!-- GENERATED:{name} END -->)", re.S)
        if not pattern.search(text):
            raise SystemExit(f"marker GENERATED:{name} not found")
        body = BLOCKS[name]()
        text = pattern.sub(lambda m, body=body: m.group(1) + body + m.group(2), text)
    return text


def main(check: bool) -> int:
    stale = []
    for fname, names in TARGETS.items():
        path = os.path.join(ROOT, fname)
        old = open(path).read()
        new = render(old, names)
        if new != old:
            stale.append(fname)
            if not check:
                open(path, "w").write(new)
    if check and stale:
        print("generated blocks out of date with the committed profiles:", ", ".join(stale), "(run python tools/design_tables.py)")
        return 1
    print("up to date" if not stale else "rewritten: " + ", ".join(stale))
    return 0


if __name__ == "__main__":
    sys.exit(main("--check" in sys.argv))
