"""bench.py --gpus N must really run N ranks: the launcher, the rendezvous, the frame sharding and
the counter reductions are exercised here on CPU (gloo, --dry-run: no GPU work, stand-in per-frame numbers); on a GPU box
the same code path initialises RCCL.  BASELINE.json configs[4] / SURVEY 8(e)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_spawns_two_ranks_gloo_dry_run():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                     # ONE line, from rank 0
    j = lines[0]
    assert j["n_gpus"] == 2 and j["rccl_world"] == 2     # counted by a real all-reduce
    assert j["frames_each_once"] is True and j["frames_per_step"] == 16
    assert j["steps"] == 3 and j["warmup"] == 1
    # 2 ranks x 8 frames x 40000 keypoints x 3 steps / max-over-ranks time (rank 1 is 1 % slower)
    assert j["value"] == pytest.approx(2 * 8 * 40000 * 3 / (3e-3 * 1.01) / 1e6, rel=1e-3)


def test_gpus_8_dry_run_gloo_the_size_the_driver_scales_to():
    """N = 8 (BASELINE.json configs[4]: 64 frames over 8 GPUs): eight ranks, every frame once, per-rank rows in rank order, and
    every rank pinned to its own non-empty slice of the node's CPUs (no 8-GPU box was ever reachable: this
    is the whole N = 8 path but the kernels)."""
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "0", "--backend", "gloo", "--dry-run"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["n_gpus"] == 8 and j["rccl_world"] == 8 and j["frames_each_once"] is True and j["frames_per_step"] == 64
    pr = j["per_rank"]
    assert len(pr["ms_per_frame"]) == 8 and len(pr["value"]) == 8
    assert pr["ms_per_frame"] == sorted(pr["ms_per_frame"])              # the stand-in makes rank r 1 % slower per rank
    assert all(n >= 1 for n in pr["cpus_pinned"])
    assert j["value"] == pytest.approx(8 * 8 * 40000 * 2 / (2e-3 * 1.07) / 1e6, rel=1e-3)


def test_cpu_slices_partition_the_allowed_set():
    import cef_loader
    sh = cef_loader.load_submodule("sharding")
    allowed = list(range(3, 3 + 21))
    parts = [sh.cpus_for_rank(allowed, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == allowed and all(parts) and max(map(len, parts)) - min(map(len, parts)) <= 1
    assert all(p == list(range(p[0], p[0] + len(p))) for p in parts)      # contiguous blocks
    assert [sh.cpus_for_rank([5, 9], r, 8) for r in range(4)] == [[5], [9], [5], [9]]      # fewer CPUs than ranks: shared
    with pytest.raises(ValueError):
        sh.cpus_for_rank([], 0, 1)


def test_driver_style_launch_matches():
    """The way the driver launches N > 1: torch.distributed.run around bench.py --gpus N."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                        "--backend", "gloo", "--dry-run"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["rccl_world"] == 2


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4", "--backend", "gloo", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_more_gpus_than_the_node_has_fails_loudly():
    """No GPU in the build container, one on the test box: --gpus 64 must exit non-zero before spawning anything."""
    r = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr


def test_gloo_without_dry_run_is_refused():
    r = _run(["--gpus", "1", "--backend", "gloo"])
    assert r.returncode != 0 and "dry-run" in (r.stderr + r.stdout)
