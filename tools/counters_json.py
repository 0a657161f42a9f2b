#!/usr/bin/env python3
"""Builds profiles/<tag>_counters.json, the file bench.py's `roofline.valu / lds / traffic` figures are computed from.

usage: tools/counters_json.py <pmc_sq.txt> <pmc_lds.txt> <traffic.json> <valu_rate.txt> <out.json> [<valu_mix.json>]
  pmc_sq.txt / pmc_lds.txt   tools/pmc_summary.py output of separate rocprofv3 --pmc passes over a short bench.py run
                             (SQ_INSTS_VALU ...; SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT)
  traffic.json               tools/traffic_json.py (FETCH_SIZE / WRITE_SIZE passes)
  valu_rate.txt              stdout of tools/microbench/valu_rate (cycles per wave-instruction per SIMD, by instruction)
  valu_mix.json              tools/valu_mix.py (static instruction mix of the kernels by rate class; built in the build container)
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


def main(sq_path, lds_path, traffic_path, valu_path, out_path, mix_path=None):
    sq, lds = parse_pmc(sq_path), parse_pmc(lds_path)
    frames = sq["fast_kernel"]["n"]
    valu_launch, valu_frame = {}, 0.0
    for k, v in sq.items():
        if k.startswith("at::") or k.startswith("__amd") or "SQ_INSTS_VALU" not in v:
            continue
        valu_launch[k] = round(v["SQ_INSTS_VALU"] / v["n"])
        valu_frame += v["SQ_INSTS_VALU"] / frames
    # the pyramid chain: every resize / tower launch of a frame (round 5: resize_rows_kernel<2> x 2 + <3>; before: resize_stream_kernel x 7)
    chain = [k for k in sq if (k.startswith("resize_") or k.startswith("pyramid_tower")) and "SQ_INSTS_VALU" in sq[k]]
    if chain:
        valu_launch["resize_chain"] = round(sum(sq[k]["SQ_INSTS_VALU"] for k in chain) / frames)
    lds_launch = {k: round(v["SQ_LDS_IDX_ACTIVE"] / v["n"]) for k, v in lds.items()
                  if "SQ_LDS_IDX_ACTIVE" in v and not k.startswith("at::") and not k.startswith("__amd")}
    conf_launch = {k: round(v["SQ_LDS_BANK_CONFLICT"] / v["n"]) for k, v in lds.items()
                   if "SQ_LDS_BANK_CONFLICT" in v and not k.startswith("at::") and not k.startswith("__amd")}
    tj = json.load(open(traffic_path)).get("per_kernel", {})
    traffic = {k.split("<")[0]: v["bytes_per_launch"] for k, v in tj.items()}
    # the pyramid chain per FRAME: every instantiation's launches (resize_rows_kernel<2> twice and <3> once per 8K frame)
    chain_t = [v for k, v in tj.items() if k.startswith("resize_") or k.startswith("pyramid_tower")]
    if chain_t:
        traffic["resize_chain"] = round(sum(v["bytes_per_launch"] * v["dispatches"] for v in chain_t) / frames)
    # issue rates (tools/microbench/valu_rate, round-4 format): a wave64 VALU instruction occupies its SIMD for 2 cycles
    # ("full rate": 32-bit add / sub / logic / right shifts / moves, fp32 add / mul / fma) or 4 ("half rate": everything else the
    # kernels are made of) -- cycles from s_memtime, the clock measured during the run
    from tools import valu_mix
    txt = open(valu_path).read()
    rates = valu_mix.parse_rates(valu_path)
    full = statistics.median(v for v in rates.values() if v < 3.0)
    half = statistics.median(v for v in rates.values() if 3.0 <= v < 6.0)
    sus = [float(x) for x in re.findall(r"\[\s*\d+\]\s+(\d+) MHz", txt)]
    clock = round(statistics.median(sus) * 1e-3, 3) if sus else float(re.search(r"attribute clock ([\d.]+) GHz", txt).group(1))
    mix = {}
    if mix_path and os.path.exists(mix_path):
        mk = json.load(open(mix_path))["kernels"]
        for name, inst in (("fast_kernel", "fast_kernel"), ("harris_kernel", "harris_kernel<1>"), ("nms_kernel", "nms_kernel<1>"),
                           ("bad_det_kernel", "bad_det_kernel"), ("bad_raw_kernel", "bad_raw_kernel<8>"), ("blur_levels_kernel", "blur_levels_kernel<false>"),
                           ("resize_chain", "resize_rows_kernel<2>"), ("resize_rows_kernel", "resize_rows_kernel<2>"), ("resize_stream_kernel", "resize_stream_kernel"),
                           ("select_kernel", "select_kernel"), ("emit_kernel", "emit_kernel"), ("angle_kernel", "angle_kernel<false>"),
                           ("angle_tail_kernel", "angle_tail_kernel")):
            if inst in mk:
                mix[name] = mk[inst]["mix_cycles_per_wave_instr"]
    res = {"note": __doc__.split("\n")[0], "git_head": os.environ.get("EFX_GIT_HEAD", "unknown (set EFX_GIT_HEAD)"),
           "source_sha256": source_digests(), "frames_profiled": frames, "clock_ghz": clock,
           "clock_from": "s_memtime / s_memrealtime during ~1 s of dense VALU work (valu_rate's sustained leg)" if sus else "device attribute",
           "cycles_full_rate": round(full, 3), "cycles_half_rate": round(half, 3),
           "cycles_per_wave_instr": round(half, 3), "cycles_per_wave_instr_by_instruction": rates,
           # three ceilings, in wave-instructions per second over the chip's 1024 SIMDs:
           "issue_peak_guide": 1024 * 2.4e9 / 2.0,            # MI355X_MICROARCH.md: wave64 VALU = 2 cycles at 2.4 GHz (every instruction full rate)
           "issue_peak_full_rate": 1024 * clock * 1e9 / full,   # measured, a kernel made of full-rate instructions only
           "issue_peak_half_rate": 1024 * clock * 1e9 / half,   # measured, a kernel made of half-rate instructions only
           "issue_peak_wave_instr_per_s": 1024 * clock * 1e9 / half,
           "mix_cycles_per_wave_instr": mix,
           "mix_from": (os.path.relpath(mix_path, ROOT) + " (tools/valu_mix.py: STATIC instruction mix of the compiled kernels by rate class)") if mix else None,
           "valu_wave_instr_per_launch": valu_launch, "valu_wave_instr_per_frame": round(valu_frame),
           "lds_cycles_per_launch": lds_launch, "lds_bank_conflict_cycles_per_launch": conf_launch,
           "traffic_bytes_per_launch": traffic}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("frames_profiled", "clock_ghz", "cycles_per_wave_instr", "valu_wave_instr_per_frame")}))


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main(*sys.argv[1:7])
