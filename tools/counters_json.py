#!/usr/bin/env python3
"""Builds profiles/<tag>_counters.json, the file bench.py's `roofline.valu / lds / traffic` figures are computed from.

usage: tools/counters_json.py <pmc_sq.txt> <pmc_lds.txt> <traffic.json> <valu_rate.txt> <out.json>
  pmc_sq.txt / pmc_lds.txt   tools/pmc_summary.py output of separate rocprofv3 --pmc passes over a short bench.py run
                             (SQ_INSTS_VALU ...; SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT)
  traffic.json               tools/traffic_json.py (FETCH_SIZE / WRITE_SIZE passes)
  valu_rate.txt              stdout of tools/microbench/valu_rate (cycles per wave-instruction per SIMD)
Per-launch figures are totals / dispatches; per-frame figures use fast_kernel's dispatch count (one per frame)."""
import hashlib
import json
import os
import re
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources of the kernels the counters describe: bench.py refuses the counters when these differ from the running tree
HEADLINE_SOURCES = ["detect_kernels.hip", "bad_kernel.hip", "blur_window.h", "bad_affine.h", "efx_device.h"]


def source_digests(root=ROOT):
    d = {}
    for f in HEADLINE_SOURCES:
        path = os.path.join(root, "cuda-efficient-features_amd", "csrc", f)
        d[f] = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
    return d


def parse_pmc(path):
    d, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*) \(dispatches (\d+)\)", line)
        if m:
            cur = m.group(1).split("<")[0]
            d.setdefault(cur, {"n": 0})
            d[cur]["n"] += int(m.group(2)) if "n_seen" not in d[cur] else 0
            d[cur]["n_seen"] = True
            continue
        m = re.match(r"\s+(\S+)\s+(\d+)\s+per-dispatch", line)
        if m and cur:
            d[cur][m.group(1)] = d[cur].get(m.group(1), 0) + int(m.group(2))
    return d


def main(sq_path, lds_path, traffic_path, valu_path, out_path):
    sq, lds = parse_pmc(sq_path), parse_pmc(lds_path)
    frames = sq["fast_kernel"]["n"]
    valu_launch, valu_frame = {}, 0.0
    for k, v in sq.items():
        if k.startswith("at::") or k.startswith("__amd") or "SQ_INSTS_VALU" not in v:
            continue
        valu_launch[k] = round(v["SQ_INSTS_VALU"] / v["n"])
        valu_frame += v["SQ_INSTS_VALU"] / frames
    if "resize_stream_kernel" in sq:
        valu_launch["resize_chain"] = round(sq["resize_stream_kernel"]["SQ_INSTS_VALU"] / frames)
    lds_launch = {k: round(v["SQ_LDS_IDX_ACTIVE"] / v["n"]) for k, v in lds.items()
                  if "SQ_LDS_IDX_ACTIVE" in v and not k.startswith("at::") and not k.startswith("__amd")}
    conf_launch = {k: round(v["SQ_LDS_BANK_CONFLICT"] / v["n"]) for k, v in lds.items()
                   if "SQ_LDS_BANK_CONFLICT" in v and not k.startswith("at::") and not k.startswith("__amd")}
    traffic = {k.split("<")[0]: v["bytes_per_launch"] for k, v in json.load(open(traffic_path)).get("per_kernel", {}).items()}
    txt = open(valu_path).read()
    clock = float(re.search(r"clock ([\d.]+) GHz", txt).group(1))
    cyc = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(k_\w+)\s+[\d.]+ us\s+->\s+([\d.]+) cycles", txt, re.M)}
    # the single-rate 32-bit / packed-16 integer instructions the detector kernels are made of (not v_pk_fma_f32, not v_mul_lo)
    # (nor the round-3 probes: k_cndmask reads a never-written VCC, the k_cmp_* pairs are two instructions per slot)
    ref = statistics.median(v for k, v in cyc.items() if k not in ("k_pk_fma_f32", "k_mullo", "k_cndmask", "k_cmp_cnd", "k_cmp_addc"))
    res = {"note": __doc__.split("\n")[0], "git_head": os.environ.get("EFX_GIT_HEAD", "unknown (set EFX_GIT_HEAD)"),
           "source_sha256": source_digests(), "frames_profiled": frames, "clock_ghz": clock,
           "cycles_per_wave_instr": round(ref, 3), "cycles_per_wave_instr_by_instruction": cyc,
           "issue_peak_wave_instr_per_s": 1024 * clock * 1e9 / ref,
           "valu_wave_instr_per_launch": valu_launch, "valu_wave_instr_per_frame": round(valu_frame),
           "lds_cycles_per_launch": lds_launch, "lds_bank_conflict_cycles_per_launch": conf_launch,
           "traffic_bytes_per_launch": traffic}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("frames_profiled", "clock_ghz", "cycles_per_wave_instr", "valu_wave_instr_per_frame")}))


if __name__ == "__main__":
    main(*sys.argv[1:6])
