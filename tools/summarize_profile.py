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

round_tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
needle = sys.argv[2] if len(sys.argv) > 2 else "step_kernel_k<skr::bf16_t, 4, true, 1, false>"  # headline: K=4 bf16 + Philox


def counter(path: str, name: str) -> list[float]:
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"] and r["Counter_Name"] == name]


rows = list(csv.DictReader(open("gpurun_out/prof_trace/r01_kernel_stats.csv")))
with open(f"profiles/{round_tag}_kernel_stats.csv", "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        wr.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

fetch = counter("gpurun_out/prof_fetch/r01_counter_collection.csv", "FETCH_SIZE")
write = counter("gpurun_out/prof_write/r01_counter_collection.csv", "WRITE_SIZE")
# steady-state launches only (the first, order-1 step of every schedule reads no history)
fetch = [v for v in fetch if v > 0.75 * max(fetch)]
out = {
    "kernel": "skr::step_kernel_k<bf16_t, K=4, NOISE=true, UV=1> (DPM-2 SDE, B=256x4x128x128)",
    "fetch_size_kib_mean": statistics.mean(fetch),
    "write_size_kib_mean": statistics.mean(write),
    "launches": [len(fetch), len(write)],
    "read_bytes_per_launch": 2 * statistics.mean(fetch) * 1024,
    "write_bytes_per_launch": statistics.mean(write) * 1024,
    "hbm_bytes_per_launch": 2 * statistics.mean(fetch) * 1024 + statistics.mean(write) * 1024,
    "algorithmic_bytes_per_launch": 256 * 4 * 128 * 128 * 10,
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B on wide coalesced streams), WRITE_SIZE x1, KiB -> bytes",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline (one counter per pass)",
}
json.dump(out, open(f"profiles/{round_tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
