#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 counter passes (one directory per pass) + the kernel-trace durations of the same runs.

  tools/summarize_counters.py profiles/r03_k2_counters.json step_kernel_k2 gpurun_out/prof_plans_*      (name filter, pass dirs)
Counters are summed over the launch by rocprofv3 and averaged here over the launches of each (kernel, grid).  HBM bytes follow
guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of wide coalesced
streams as 64 B (x2), WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections, csv, glob, json, os, statistics, sys

out_path, needle, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
counters: dict = collections.defaultdict(lambda: collections.defaultdict(list))
durations: dict = collections.defaultdict(list)


def short(name: str) -> str:
    return name.replace("skr::", "").replace("void ", "")[:150]


for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if needle in r["Kernel_Name"]:
                counters[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        if not os.path.basename(os.path.normpath(d)).endswith("_trace"):
            continue  # durations only from the plain trace pass (counter passes serialise and slow the kernels)
        for r in csv.DictReader(open(path)):
            if needle in r["Kernel_Name"]:
                grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                durations[(short(r["Kernel_Name"]), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

result = {}
for key in sorted(set(counters) | set(durations)):
    name, grid = key
    c = {k: statistics.mean(v) for k, v in counters.get(key, {}).items()}
    n = {k: len(v) for k, v in counters.get(key, {}).items()}
    entry: dict = {"grid_threads": grid, "counters_mean_per_launch": c, "launches_per_counter": n}
    if key in durations:
        d = durations[key]
        entry["duration_ns"] = {"calls": len(d), "mean": statistics.mean(d), "min": min(d), "max": max(d)}
    dv = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        dv["hbm_read_bytes"] = 2 * c["FETCH_SIZE"] * 1024
        dv["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        dv["hbm_bytes"] = dv["hbm_read_bytes"] + dv["hbm_write_bytes"]
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
            if k in c:
                dv[k.lower() + "_frac_of_wave_cycles"] = c[k] / wc
    if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
        dv["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if c.get("SQ_WAVES") and "SQ_INSTS_LDS" in c:
        dv["lds_instructions_per_wave"] = c["SQ_INSTS_LDS"] / c["SQ_WAVES"]
    if c.get("SQ_WAVES") and wc:
        dv["wave_lifetime_quad_cycles"] = wc / c["SQ_WAVES"]
    if "TCC_HIT_sum" in c and c.get("TCC_MISS_sum"):
        dv["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    entry["derived"] = dv
    result[f"{name} @grid {grid}"] = entry
json.dump({"filter": needle, "passes": dirs, "kernels": result,
           "note": "SQ_* cycle counters are quad-cycles summed over waves; FETCH_SIZE x2 and KiB -> bytes per guides/MI355X_MICROARCH.md"}, open(out_path, "w"), indent=1)
print(json.dumps({k: v["derived"] | ({"mean_us": v["duration_ns"]["mean"] / 1e3} if "duration_ns" in v else {}) for k, v in result.items()}, indent=1))
