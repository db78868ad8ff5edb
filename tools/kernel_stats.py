#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace output directory into one small CSV: a row per (kernel, workgroups, workgroup size) with
calls / total / average / min / max / stddev in ns -- rocprofv3's own --stats table lumps launches of one kernel at different
grid sizes together (bench.py runs the headline kernel at B=256 and, in its graph-loop key, at B=64).

usage: tools/kernel_stats.py <rocprofv3 output dir> <out.csv>
"""
import collections
import functools
import csv
import glob
import os
import shutil
import statistics
import subprocess
import sys


@functools.lru_cache(maxsize=None)
def demangle(name: str) -> str:
    """rocprofv3 leaves kernels whose signature holds a __bf16 mangled (`DF16b` is newer than its demangler): substitute the
    vendor-type spelling GNU c++filt understands and demangle; anything else is returned as it came"""
    if not name.startswith("_Z"):
        return name
    tool = shutil.which("c++filt")
    if tool is None:
        return name
    out = subprocess.run([tool, name.replace("DF16b", "u6__bf16")], capture_output=True, text=True).stdout.strip()
    return out.replace("__bf16", "skr::bf16_t") if out and not out.startswith("_Z") else name


def main(raw: str, out: str) -> None:
    files = glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {raw}")
    groups: dict = collections.defaultdict(list)
    for path in files:
        for r in csv.DictReader(open(path)):
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            groups[(demangle(r["Kernel_Name"]), grid // wg, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in groups.values())
    with open(out, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["Name", "Workgroups", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for (name, wgs, wg), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            wr.writerow([name[:200], wgs, wg, len(d), sum(d), f"{statistics.mean(d):.3f}", f"{100 * sum(d) / total:.2f}", min(d), max(d), f"{statistics.pstdev(d):.3f}"])
    print(f"{out}: {len(groups)} rows")


def redo(path: str) -> None:
    "demangle the Name column of an already condensed CSV in place"
    rows = list(csv.reader(open(path)))
    for r in rows[1:]:
        r[0] = demangle(r[0])[:200]
    csv.writer(open(path, "w", newline="")).writerows(rows)


if __name__ == "__main__":
    if sys.argv[1] == "--demangle":
        for p in sys.argv[2:]:
            redo(p)
    else:
        main(sys.argv[1], sys.argv[2])
