"""The driver's contract for bench.py, on the GPU: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with
BASELINE.json's metric, the `roofline` and `cpu_baseline` objects and (round 4) the `configs` block, and the product path it
times equals the oracle on frame 0 (`parity_8k_frame0`).  A short run: the kernels are the same as in the default command."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--config-iters", "3"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0]


def test_contract_fields(line):
    j = line
    assert j["metric"].startswith("Mkeypoints/s detectAndCompute (8K, 40k kp, BAD512)") and j["unit"] == "Mkeypoints/s"
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["dtype"] == "u8" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 20.0 and j["value"] == pytest.approx(8 * j["config"]["keypoints_per_frame"] / (j["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    assert j["vs_baseline"] == pytest.approx(j["value"] / (40000 / 8.2e-3 / 1e6), rel=1e-2)
    assert j["parity_8k_frame0"] is True                          # the timed path == the oracle on the bench frame
    # 6 steps are over in 16 ms: the same steps run again for --sustain-seconds (default 2 s) and are reported beside `value`
    s = j["sustained"]
    assert s["seconds"] >= 1.7 and s["steps"] > 100 and s["value"] == pytest.approx(s["steps"] * 8 * j["config"]["keypoints_per_frame"] / s["seconds"] / 1e6, rel=1e-3)
    assert 0.5 * j["value"] < s["value"] < 1.5 * j["value"]


def test_roofline_and_cpu_baseline(line):
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["kernel"] in r["kernels_isolated"] and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-2)
    # achieved = algorithmic bytes per launch / the kernel's average launch duration (HIP events on the launch stream)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-2)
    assert {"fast_kernel", "harris_kernel", "nms_kernel", "blur_levels_kernel", "bad_raw_kernel"} <= set(r["kernels_isolated"])
    assert "pyramid_fast" in r and 0 < r["pyramid_fast"]["frac"] < 1
    c = r["counters"]
    if c["sources_match_this_tree"]:                              # PMC-derived parts only from counters of THIS tree's kernels
        v = r["valu"]
        assert v["peaks"]["guide"] == pytest.approx(1228.8) and v["peaks"]["half_rate_measured"] < v["peaks"]["full_rate_measured"]
        w = v["whole_frame"]
        assert 0 < w["frac_vs_guide_peak"] < w["frac_vs_mix_ceiling"] < w["frac_vs_half_rate_ceiling"] < 1.2
        assert r["traffic"] is not None
    else:
        assert r["valu"] is None and r["traffic"] is None and "why_null" in c
    b = line["cpu_baseline"]
    assert b["kind"] == "port" and b["cores"] == 1 and b["unit"] == "Mkeypoints/s" and b["repeats"] >= 3 and 0 < b["value"] < 1


def test_configs_block(line):
    rows = line["configs"]["rows"]
    by = {}
    for r in rows:
        by.setdefault(r["config"], []).append(r)
    assert len(by["C2"]) == 1 and len(by["C3"]) == 2 and len(by["C4"]) == 2
    assert by["C3"][0]["keypoints"] == 40000 and by["C4"][1]["keypoints"] == 40000
    for r in by["C2"] + by["C3"]:
        assert r["roofline"]["survey_8d_MB"] > 0 and r["roofline"]["design_MB"] > 0 and r["cpu_baseline"]["repeats"] >= 3
    sizes = {(r.get("mode"), r.get("size"), r.get("descriptor")) for r in by["readme"]}
    assert ("detect", "8k", None) in sizes and ("compute + detectAndCompute", "8k", "BAD512") in sizes
    # frame-batched launches (round 6): FHD x 16 and 4K x 8 per launch chain, beside the per-frame form of the same entry point
    b = {r["size"]: r for r in by["batch"]}
    assert b["fhd"]["frames_per_launch"] == 16 and b["4k"]["frames_per_launch"] == 8
    assert b["fhd"]["batched"]["frames_per_s"] > b["fhd"]["per_frame_calls"]["frames_per_s"] > 1000
    # dense frames are complete on a context's first call
    for r in by["data"]:
        assert r["first_call_keypoints"] == r["keypoints"] == 40000 and r["overflow_events"] == 0


def test_force_dist_runs_rccl_at_world_1():
    """SCALE keeps being skipped, so the N > 1 branch of bench.py has never run with RCCL in a recorded run.
    `--force-dist` initialises the nccl (= RCCL) process group at world size 1 and sends the same counter reductions / gathers
    through it on device tensors; `rccl_world` then comes from a real all-reduce.  (No scaling curve is claimed from this.)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--force-dist",
                        "--no-configs", "--no-cpu-baseline", "--sustain-seconds", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert j["rccl_exercised"] is True and j["rccl_world"] == 1 and j["n_gpus"] == 1
    assert j["value"] > 20.0 and len(j["per_rank"]["ms_per_frame"]) == 1
