#!/usr/bin/env python3
"""GPU busy time vs span of a rocprofv3 kernel trace: sum of kernel durations, idle gaps between consecutive kernels (by start time),
per kernel-name totals.  usage: tools/trace_gaps.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy, gaps, end = 0, [], None
names = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None:
        if s > end: gaps.append(s - end)
        s_eff = max(s, end)
    else:
        s_eff = s
    busy += max(0, e - s_eff)
    end = e if end is None else max(end, e)
    n = r["Kernel_Name"].replace("skr::", "").replace("void ", "")[:70]
    names[n][0] += 1; names[n][1] += e - s
span = t1 - t0
print(f"kernels {len(rows)}  span {span / 1e3:.1f} us  busy {busy / 1e3:.1f} us ({busy / span:.3f})  gaps: {len(gaps)} totalling {sum(gaps) / 1e3:.1f} us, median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us, max {max(gaps) / 1e3 if gaps else 0:.1f} us")
for n, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {n:70s} {c:5d} calls {d / 1e3:10.1f} us")
