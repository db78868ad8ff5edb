#!/usr/bin/env python3
"""Static instruction mix of the generator kernels, priced with the measured issue costs (profiles/r04_valu_rates.txt):
compiles a csrc file for gfx950 with the library's own flags (--save-temps), cuts each kernel out of the assembly and counts
its VALU / LDS / VMEM instructions by cost class (full rate 1, half rate 1.75: packed fp32, 32-bit integer multiplies and
v_mad_u64_u32, conversions, fp64 adds, v_lshl_add; quarter rate 3.4: the transcendentals).  Static counts, not executed
counts: loops count once -- what the table gives is the MIX (cost-weighted slots per instruction) that turns the executed
instruction count of the SQ counters into issue slots.

usage: tools/isa_mix.py skrample_amd/csrc/skr_colored.hip out.json [kernel-name-substring ...]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HALF = ("v_pk_", "v_mad_u64", "v_mad_i64", "v_lshl_add", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_", "v_bfrev", "v_mul_lo", "v_mul_hi", "v_add_lshl", "v_and_or", "v_lshl_or", "v_add3", "v_rndne", "v_fract", "v_ldexp", "v_frexp")
QUARTER = ("v_log_f32", "v_exp_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_rsq_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_iflag")


def cost(op: str) -> float:
    return 3.4 if op.startswith(QUARTER) else 1.75 if op.startswith(HALF) else 1.0


def main(src: str, out: str, needles: list[str]) -> None:
    import __graft_entry__ as G

    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *G.PER_FILE_FLAGS.get(os.path.basename(src), [])]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["hipcc", *flags, "--save-temps", "-c", "-o", os.path.join(tmp, "x.o"), os.path.abspath(src)], check=True, cwd=tmp, capture_output=True)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        text = open(os.path.join(tmp, asm)).read()
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", text, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1).replace("DF16b", "u6__bf16")], capture_output=True, text=True).stdout.strip()
        if needles and not any(n in name for n in needles):
            continue
        ops = [l.split()[0] for l in m.group(0).splitlines() if re.match(r"\s+(v_|ds_|global_|buffer_|s_barrier)", l)]
        valu = [o for o in ops if o.startswith("v_")]
        slots = sum(cost(o) for o in valu)
        kernels[name[:140]] = {
            "valu_instructions_static": len(valu),
            "valu_issue_slots_static": round(slots, 1),
            "slots_per_instruction": round(slots / max(len(valu), 1), 4),
            "half_rate_instructions": sum(1 for o in valu if cost(o) == 1.75),
            "quarter_rate_instructions": sum(1 for o in valu if cost(o) == 3.4),
            "lds_instructions_static": sum(1 for o in ops if o.startswith("ds_")),
            "vmem_instructions_static": sum(1 for o in ops if o.startswith(("global_", "buffer_"))),
            "barriers_static": sum(1 for o in ops if o == "s_barrier"),
        }
    json.dump({"source": src, "flags": flags, "costs": {"full": 1.0, "half": 1.75, "quarter": 3.4}, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k[:90]:90s} VALU {v['valu_instructions_static']:5d}  slots {v['valu_issue_slots_static']:7.1f}  ({v['slots_per_instruction']:.3f}/instr)  LDS {v['lds_instructions_static']}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
