#!/usr/bin/env python3
"""The VALU / LDS roofline of the VALU-bound noise generators: the minimum issue slots and LDS
cycles per unit of work, next to what the SQ counters measured, for the Colored plane kernels at 128 x 128 and the Pyramid main pass
at 256 x 256.  Everything is computed from committed files:

  profiles/r04_valu_rates.txt            issue cost per instruction class, measured (tools/tune/valu_rates.hip)
  profiles/r04_colored_sq_counters.json  SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_WAVES per launch and the launch durations (tools/collect_r04.sh colored)
  profiles/r04_pyramid_sq_counters.json  the same for pyramid_pass1
  profiles/r04_colored_isa_mix.json      static instruction mix of the kernels (tools/isa_mix.py): slots per executed instruction
  profiles/r04_colored_rng_share.txt     the forward kernel timed with the draw replaced by a hash (tools/tune/tune_colored.hip -DSKR_STUB_DRAW)

usage: python tools/generator_roofline.py  ->  profiles/r04_colored_valu_roofline.json (also printed)

One ISSUE SLOT = one full-rate wave instruction = 4 shader cycles of a SIMD.  Costs (measured): plain fp32 / integer / logic 1 slot;
v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32, v_mad_u64_u32, v_mul_lo/hi_u32, v_lshl_add_u32, v_cvt_*, v_add_f64 1.75 slots;
v_log / v_exp / v_sqrt / v_sin / v_cos / v_rcp 3.4 slots.  (So a packed fp32 instruction buys 4 flops for 1.75 slots = 2.29 flops per
slot against 2 for v_fma_f32: the "4 flops per v_pk_fma_f32" of the review would be a 2x gain, the chip gives 14 %.)
LDS, per CU: ds_read_b64 3.5 cycles per wave instruction, ds_write_b64 10.4 (a write also carries its data registers to the LDS)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
FULL, HALF, QUARTER = 1.0, 1.75, 3.4
LDS_READ_B64, LDS_WRITE_B64 = 3.5, 10.4  # CU-wide cycles per wave instruction (rates file: 8.02 / 24.0 ticks per SIMD stream, 4 SIMDs, 1.74 cycles per tick)

# one Philox4x32-10 block + two Box-Muller pairs = 4 normals, as the compiler emits it for a per-sample-uniform key and stream
# (counted from the assembly of skr_philox.h::normal4: the first two rounds are half scalar)
NORMAL4 = {"v_mad_u64_u32": (18, HALF), "v_xor_b32": (34, FULL), "v_mov_b32": (4, FULL), "v_cvt_f32_u32": (4, HALF), "v_fma_f32 (u01)": (4, FULL),
           "v_log/v_sqrt/v_sin/v_cos": (8, QUARTER), "v_mul_f32": (6, FULL)}  # fmt: skip


def slots(table: dict) -> float:
    return sum(n * c for n, c in table.values())


def counters(path: str, needle: str) -> dict:
    d = json.load(open(os.path.join(P, path)))
    for name, v in d["kernels"].items():
        if needle in name:
            c = v["counters_mean_per_launch"]
            return {"name": name, "waves": c["SQ_WAVES"], "valu_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"], "lds_per_wave": c["SQ_INSTS_LDS"] / c["SQ_WAVES"],
                    "bank_conflict_share": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), "duration_us": v["duration_ns"]["mean"] / 1e3 if "duration_ns" in v else None}  # fmt: skip
    raise KeyError(needle)


def mix(path: str, needle: str) -> float:
    d = json.load(open(os.path.join(P, path)))
    for name, v in d["kernels"].items():
        if needle in name:
            return v["slots_per_instruction"]
    raise KeyError(needle)


def main() -> None:
    out: dict = {"unit": "issue slot = one full-rate wave instruction = 4 SIMD cycles", "costs": {"full": FULL, "half": HALF, "quarter": QUARTER,
                 "lds_read_b64_cu_cycles": LDS_READ_B64, "lds_write_b64_cu_cycles": LDS_WRITE_B64, "source": "profiles/r04_valu_rates.txt"}}  # fmt: skip
    n4 = slots(NORMAL4)
    out["normal4"] = {"instructions": sum(n for n, _ in NORMAL4.values()), "issue_slots": n4, "mix": {k: {"count": n, "cost": c} for k, (n, c) in NORMAL4.items()}}
    H = W = 128
    threads, waves = 512, 8
    line_flops = 5 * W * 7  # 5 N log2 N
    # ---- forward plane kernel: draw, 64 row-pair transforms, untangle, 65 column transforms, half spectrum out
    fwd_flops = (H // 2 + W // 2 + 1) * line_flops + (H // 2) * (W // 2 + 1) * 8
    fwd = {
        "draw_slots": (H * W // 4) * n4 / threads,
        "transform_slots_at_2_flops_per_slot": fwd_flops / threads / 2.0,
        "transform_slots_at_2.29_flops_per_slot": fwd_flops / threads / (4 / HALF),
        "lds_round_trips": {"reads": 6, "writes": 6, "note": "draw write; rows 8-point + two radix-4 passes; untangle (read + write); columns: one radix-4 pass in LDS, the last one stores to HBM"},
    }
    tile = (W // 2 + 1) * (H + 1)  # complex points of the (larger) column tile
    fwd["lds_cu_cycles_per_plane"] = (tile / 64) * (6 * LDS_READ_B64 + 6 * LDS_WRITE_B64)
    fwd["min_valu_slots_per_wave"] = fwd["draw_slots"] + fwd["transform_slots_at_2.29_flops_per_slot"]
    fwd["min_valu_simd_cycles_per_plane"] = fwd["min_valu_slots_per_wave"] * 4 * waves / 4  # 8 waves on 4 SIMDs
    m = counters(f"r04_colored_sq_counters.json", "colored_plane<0, float, 7, 7>")
    spi = mix("r04_colored_isa_mix.json", "colored_plane<0, float, 7, 7>")
    fwd["measured"] = {**m, "slots_per_instruction_static_mix": spi, "valu_slots_per_wave": m["valu_per_wave"] * spi,
                       "measured_over_minimum": m["valu_per_wave"] * spi / fwd["min_valu_slots_per_wave"]}  # fmt: skip
    out["colored_plane<0> (forward, 128x128)"] = fwd
    # ---- inverse plane kernel: no draw; spectrum in, 65 column + 64 row-pair transforms, packing, scaled rounded result out
    inv_flops = (H // 2 + W // 2 + 1) * line_flops + (H // 2) * W * 4 + H * W * 2
    inv = {"transform_slots_at_2.29_flops_per_slot": inv_flops / threads / (4 / HALF), "lds_round_trips": {"reads": 6, "writes": 5}}
    inv["lds_cu_cycles_per_plane"] = (tile / 64) * (6 * LDS_READ_B64 + 5 * LDS_WRITE_B64)
    inv["min_valu_slots_per_wave"] = inv["transform_slots_at_2.29_flops_per_slot"]
    m = counters("r04_colored_sq_counters.json", "colored_planeILi1E")
    spi = mix("r04_colored_isa_mix.json", "colored_plane<1")
    inv["measured"] = {**m, "slots_per_instruction_static_mix": spi, "valu_slots_per_wave": m["valu_per_wave"] * spi,
                       "measured_over_minimum": m["valu_per_wave"] * spi / inv["min_valu_slots_per_wave"],
                       "note": "not VALU-bound: its blocks wait 4.4 us of their 14 us for the spectrum (tools/tune/tune_colored.hip timeline), two 67 KiB blocks per CU cannot cover it"}  # fmt: skip
    out["colored_plane<1> (inverse, 128x128)"] = inv
    # ---- what the forward kernel spends on the generator, measured by removing it
    share = os.path.join(P, "r04_colored_rng_share.txt")
    if os.path.isfile(share):
        nums = [float(x) for x in re.findall(r"forward kernel: first start -> last store issue ([0-9.]+) us", open(share).read())]
        if len(nums) == 2:
            out["rng_share_of_the_forward_kernel"] = {"with_philox_us": nums[0], "draw_stubbed_us": nums[1], "share": 1 - nums[1] / nums[0], "source": "profiles/r04_colored_rng_share.txt"}
    # ---- Pyramid main pass at 64 x (4, 256, 256): two full-resolution normals per pixel (base + level 0) cannot be avoided
    try:
        m = counters("r04_pyramid_sq_counters.json", "pyramid_pass1<true, 1024>")  # (the 256 x 256 planes of cfg5 take the 1024-lane strip kernel)
        spi = mix("r04_pyramid_isa_mix.json", "pyramid_pass1<true, 1024>")
        px_per_thread = 256 * 256 / 1024
        draw = 2 * (px_per_thread / 4) * n4
        out["pyramid_pass1 (256x256 plane per block)"] = {
            "draw_slots_per_wave": draw, "note": "the level blends (1 FMA per pixel and level with the interpolated coarse rows in registers) and the statistics add ~8 slots per pixel",
            "min_valu_slots_per_wave": draw + px_per_thread * 8,
            "measured": {**m, "slots_per_instruction_static_mix": spi, "valu_slots_per_wave": m["valu_per_wave"] * spi, "measured_over_minimum": m["valu_per_wave"] * spi / (draw + px_per_thread * 8)},
        }
    except (KeyError, FileNotFoundError):
        pass
    json.dump(out, open(os.path.join(P, "r04_colored_valu_roofline.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
