#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the small, committed files under profiles/.

  profiles/<round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (kernel names shortened)
  profiles/<round>_pmc_traffic.json   HBM bytes per launch of the headline kernel from the FETCH_SIZE and
                                      WRITE_SIZE passes, corrected as guides/MI355X_MICROARCH.md (HBM section)
                                      prescribes: values are KiB; on gfx950 FETCH_SIZE counts 128-B requests as
                                      64 B for wide coalesced streams -> x2; WRITE_SIZE is exact.
usage: tools/summarize_profile.py r01 [kernel-name-substring]
"""
import csv
import json
import statistics
import sys

import os

RAW = os.environ.get("SKR_PROF_RAW", "gpurun_out")   # where the rocprofv3 pass directories (prof_trace, prof_fetch, ...) are
OUT = os.environ.get("SKR_PROF_OUT", "profiles")     # where the summaries go
round_tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
needle = sys.argv[2] if len(sys.argv) > 2 else "step_kernel_k1<skr::bf16_t, 4, true, false, true, false>"  # headline: K=4 bf16 + Philox, one-trip paced kernel, kernarg scalars
HEADLINE_GRID = 256 * 4 * 128 * 128 // 8  # threads of a B=256 launch: bench.py's extra graph-loop key runs the same kernel at B=64


def counter(path: str, name: str) -> list[float]:
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"] and r["Counter_Name"] == name and int(r["Grid_Size"]) == HEADLINE_GRID]


# kernel stats rebuilt from the per-dispatch trace, one row per (kernel, grid size): the same kernel runs at B=256 (the timed
# headline launches) and at B=64 (the graph-loop key of the same bench command), which rocprofv3's own --stats table lumps together
import collections

groups: dict = collections.defaultdict(list)
for r in csv.DictReader(open(f"{RAW}/prof_trace/prof_kernel_trace.csv")):
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    groups[(r["Kernel_Name"], grid // wg, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in groups.values())
with open(f"{OUT}/{round_tag}_kernel_stats.csv", "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["Name", "Workgroups", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for (name, wgs, wg), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        wr.writerow([name[:160], wgs, wg, len(d), sum(d), f"{statistics.mean(d):.3f}", f"{100 * sum(d) / total:.2f}", min(d), max(d), f"{statistics.pstdev(d):.3f}"])

fetch = counter(f"{RAW}/prof_fetch/prof_counter_collection.csv", "FETCH_SIZE")
write = counter(f"{RAW}/prof_write/prof_counter_collection.csv", "WRITE_SIZE")
# steady-state launches only (the first, order-1 step of every schedule reads no history)
fetch = [v for v in fetch if v > 0.75 * max(fetch)]
out = {
    "kernel": "skr::step_kernel_k1<bf16_t, K=4, NOISE=true> (DPM-2 SDE, B=256x4x128x128)",
    "fetch_size_kib_mean": statistics.mean(fetch),
    "write_size_kib_mean": statistics.mean(write),
    "launches": [len(fetch), len(write)],
    "read_bytes_per_launch": 2 * statistics.mean(fetch) * 1024,
    "write_bytes_per_launch": statistics.mean(write) * 1024,
    "hbm_bytes_per_launch": 2 * statistics.mean(fetch) * 1024 + statistics.mean(write) * 1024,
    "algorithmic_bytes_per_launch": 256 * 4 * 128 * 128 * 10,
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B on wide coalesced streams), WRITE_SIZE x1, KiB -> bytes",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline (one counter per pass; B=256 launches only)",
}
json.dump(out, open(f"{OUT}/{round_tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))


# SQ-level counters of the headline kernel (one or more passes; every counter summed over the launch, mean over launches)
import glob
import os

sq = {}
for path in sorted(glob.glob(f"{RAW}/prof_sq*/prof_counter_collection.csv")) + sorted(glob.glob(f"{RAW}/prof_misc/prof_counter_collection.csv")):
    acc: dict[str, list[float]] = {}
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"] and int(r["Grid_Size"]) == HEADLINE_GRID:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        sq[k] = {"mean_per_launch": statistics.mean(v), "launches": len(v)}
if sq:
    g = lambda k: sq.get(k, {}).get("mean_per_launch")
    derived = {}
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
        wc = g("SQ_WAVE_CYCLES")
        derived["wait_any_frac_of_wave_cycles"] = g("SQ_WAIT_ANY") / wc
        if g("SQ_ACTIVE_INST_ANY") is not None:
            derived["active_inst_any_frac_of_wave_cycles"] = g("SQ_ACTIVE_INST_ANY") / wc
        if g("SQ_WAIT_INST_ANY") is not None:
            derived["wait_inst_any_frac_of_wave_cycles"] = g("SQ_WAIT_INST_ANY") / wc
        if g("SQ_ACTIVE_INST_VALU") is not None:
            derived["active_inst_valu_frac_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / wc
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        derived["valu_instructions_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    if g("SQ_WAVES") and g("SQ_WAVE_CYCLES"):
        derived["wave_lifetime_quad_cycles"] = g("SQ_WAVE_CYCLES") / g("SQ_WAVES")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
        derived["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    json.dump({"kernel": needle, "counters": sq, "derived": derived,
               "note": "SQ_* cycle counters are in quad-cycles summed over waves (guides/MI355X_MICROARCH.md); WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES"},
              open(f"{OUT}/{round_tag}_sq_counters.json", "w"), indent=1)
    print(json.dumps(derived, indent=1))
