#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace output directory into one small CSV: a row per (kernel, workgroups, workgroup size) with
calls / total / average / min / max / stddev in ns -- rocprofv3's own --stats table lumps launches of one kernel at different
grid sizes together (bench.py runs the headline kernel at B=256 and, in its graph-loop key, at B=64).

usage: tools/kernel_stats.py <rocprofv3 output dir> <out.csv>
"""
import collections
import csv
import glob
import os
import statistics
import sys


def main(raw: str, out: str) -> None:
    files = glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {raw}")
    groups: dict = collections.defaultdict(list)
    for path in files:
        for r in csv.DictReader(open(path)):
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            groups[(r["Kernel_Name"], grid // wg, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in groups.values())
    with open(out, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["Name", "Workgroups", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for (name, wgs, wg), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            wr.writerow([name[:200], wgs, wg, len(d), sum(d), f"{statistics.mean(d):.3f}", f"{100 * sum(d) / total:.2f}", min(d), max(d), f"{statistics.pstdev(d):.3f}"])
    print(f"{out}: {len(groups)} rows")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
