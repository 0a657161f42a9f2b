"""Short determinism soaks in the GPU tier: the same frames over and over on three contexts / streams, every result compared on
the device with the first result of its frame -- the single-frame calls (tools/microbench/soak_diag.py) and the batched path
(soak_batch.py).  Seconds, not minutes: a gross cross-frame dependence (state that survives a frame in a context's buffers, a
hand-off inside a launch that does not wait) shows here; the one-in-500 000 race of docs/history/round6.md section 11 needed minutes
(tools/verify_build.sh has the longer legs)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args):
    p = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("size", ["fhd", "4k"])
def test_single_frame_calls_repeat_bit_for_bit(size):
    out = _run(["tools/microbench/soak_diag.py", "6", "3", "40000", size])
    m = re.search(r"frames (\d+) events (\d+)", out)
    assert m, out[-1000:]
    assert int(m.group(1)) > 1000 and int(m.group(2)) == 0, out[-3000:]


@pytest.mark.parametrize("size,frames,desc", [("fhd", "16", "BAD_512"), ("4k", "8", "HASH_SIFT_256")])
def test_batched_launches_repeat_bit_for_bit(size, frames, desc):
    out = _run(["tools/microbench/soak_batch.py", "6", size, frames, desc])
    m = re.search(r"frames (\d+) mismatching results per frame slot \[([0-9, ]+)\]", out)
    assert m, out[-1000:]
    assert int(m.group(1)) > 500 and all(int(x) == 0 for x in m.group(2).split(",")), out[-3000:]
