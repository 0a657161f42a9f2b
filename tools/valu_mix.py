#!/usr/bin/env python3
"""Static VALU instruction mix of the product's kernels, by ISSUE RATE class (build container, no GPU):

    python tools/valu_mix.py profiles/r04_valu_rate.txt profiles/r04_valu_mix.json

Why: on gfx950 a wave64 VALU instruction takes 2 cycles of its SIMD ("full rate": v_mov / v_add / v_sub / v_and / v_or / v_xor /
v_not / v_lshrrev / v_ashrrev _b32, v_add / v_sub / v_mul / v_fma / v_fmac _f32, v_add_f16, v_accvgpr_*) or 4 cycles ("half
rate": everything else the kernels use -- min / max, v_lshlrev, all three-operand integer forms, packed 16-bit, dot, perm,
conversions, packed fp32, fp64, SDWA / DPP forms, compares, v_cndmask, lane moves -- AND any full-rate opcode with a scalar-register
source operand) or 8 (transcendentals); measured by
tools/microbench/valu_rate (profiles/rNN_valu_rate.txt).  The SQ counters do not tell the classes apart (SQ_ACTIVE_INST_VALU ==
SQ_INSTS_VALU for both, profiles/r04_valu_calib.txt), so the issue ceiling of a kernel's MIX is estimated from its ISA: every
VALU instruction of the compiled kernel is looked up in the measured table, and the kernel's mean cycles per wave-instruction
is the count-weighted mean.  STATIC counts (an unrolled loop body counts once per copy, a loop once): an estimate of the
dynamic mix, good where the hot code is straight-line or a loop of the same make-up, which is what these kernels are.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-efficient-features_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = "-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -S".split()
SOURCES = ["detect_kernels.hip", "bad_kernel.hip", "hashsift_kernels.hip"]
NOT_VALU = ("v_mfma", "v_smfmac")            # matrix pipe: not a VALU issue slot


def parse_rates(path):
    """name -> cycles per wave-instruction at 8 waves per SIMD, from tools/microbench/valu_rate's table."""
    rates = {}
    for line in open(path):
        m = re.match(r"^(v_\S+)(?: \([^)]*\))?\s+.*?\|\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+([\d.]+)\s+\|", line)
        if m and " (" not in line.split("|")[0].strip().split("  ")[0]:
            rates.setdefault(m.group(1), float(m.group(2)))
    return rates


def base(mnemonic):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mnemonic)


def classify(mnemonic, rates, full, half, quarter, sgpr=False):
    """cycles for one static VALU instruction: its measured row when there is one, else the class of its kind.  A full-rate
    instruction with a scalar-register source runs at half rate (measured: the "(sgpr)" rows of the table)."""
    if sgpr:
        c, k = classify(mnemonic, rates, full, half, quarter)
        return (half, "half (full-rate opcode with an SGPR source)") if k == "full" else (c, k)
    b = base(mnemonic)
    if mnemonic.endswith(("_sdwa", "_dpp")):
        return half, "half"                                  # measured: v_add_u32_sdwa / _dpp run at half rate
    if b in rates and b != "v_cndmask_b32":
        c = rates[b]
        return (full, "full") if c < 3.0 else (half, "half") if c < 6.0 else (quarter, "quarter")
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", b):
        return quarter, "quarter"
    if re.match(r"v_(subrev_f32|subrev_u32|mac_f32|subb|subbrev)", b):
        return (full, "full") if b in ("v_subrev_f32", "v_subrev_u32") else (half, "half")
    return half, "half (not in the table: counted as half rate)"


def kernel_bodies(asm):
    cur, out = None, {}
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); out[cur] = []
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        if t.startswith(".") or t.startswith(";") or not t:
            continue
        # mnemonic, and whether a SOURCE operand is a scalar register (s12, s[4:5], vcc, ttmp...; not the destination of a compare)
        parts = t.split(None, 1)
        ops = parts[1].split(";")[0] if len(parts) > 1 else ""
        srcs = ops.split(",")[1:]
        sgpr = any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|m0|ttmp)", o) for o in srcs)
        out[cur].append((parts[0], sgpr))
    return out


def demangle(names):
    p = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return [re.sub(r"\(anonymous namespace\)::", "", l).replace("void ", "").split("(")[0] for l in p.stdout.strip().splitlines()]


def main(rate_path, out_path):
    rates = parse_rates(rate_path)
    import statistics
    full = statistics.median(v for v in rates.values() if v < 3.0)
    half = statistics.median(v for v in rates.values() if 3.0 <= v < 6.0)
    quarter = statistics.median(v for v in rates.values() if 6.0 <= v < 12.0)
    res = {"note": __doc__.split("\n")[0], "rates_from": os.path.relpath(rate_path, ROOT),
           "cycles_full_rate": round(full, 3), "cycles_half_rate": round(half, 3), "cycles_quarter_rate": round(quarter, 3), "kernels": {}}
    for src in SOURCES:
        asm = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, src), "-o", "-"], capture_output=True, text=True, check=True).stdout
        bodies = kernel_bodies(asm)
        names = list(bodies)
        for mangled, pretty in zip(names, demangle(names)):
            ins = [(i, sg) for i, sg in bodies[mangled] if i.startswith("v_") and not i.startswith(NOT_VALU)]
            if len(ins) < 16:
                continue
            cyc, cls, unknown, demoted = 0.0, {"full": 0, "half": 0, "quarter": 0}, {}, {}
            for i, sg in ins:
                c, k = classify(i, rates, full, half, quarter, sg)
                cyc += c
                cls[k.split(" ")[0]] += 1
                if "not in the table" in k:
                    unknown[base(i)] = unknown.get(base(i), 0) + 1
                if "SGPR source" in k:
                    demoted[base(i)] = demoted.get(base(i), 0) + 1
            res["kernels"][pretty] = {"valu_static": len(ins), "full_rate_share": round(cls["full"] / len(ins), 3),
                                      "half_rate_share": round(cls["half"] / len(ins), 3), "quarter_rate_share": round(cls["quarter"] / len(ins), 3),
                                      "mix_cycles_per_wave_instr": round(cyc / len(ins), 3),
                                      "full_rate_opcodes_with_sgpr_source": dict(sorted(demoted.items(), key=lambda kv: -kv[1])[:8]),
                                      "not_in_table": dict(sorted(unknown.items(), key=lambda kv: -kv[1])[:8])}
    json.dump(res, open(out_path, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:48s} {v['valu_static']:6d} VALU  full {v['full_rate_share']:.2f}  mix {v['mix_cycles_per_wave_instr']:.2f} cycles  sgpr-demoted {sum(v['full_rate_opcodes_with_sgpr_source'].values())}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
