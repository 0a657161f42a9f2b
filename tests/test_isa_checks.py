"""Properties of the COMPILED kernels that the source relies on and no run-time test would catch (build container, no GPU:
hipcc cross-compiles gfx950 assembly in seconds):
  * no kernel of the library uses scratch memory -- a dispatch with a private segment costs this runtime ~25 us of stall per
    launch that no kernel timer shows (DESIGN history, round 4);
  * resize_rows_kernel writes M0 inside its LDS-DMA statements without saving it (round 5: two scalar instructions per
    source row less): nothing the compiler emits in that kernel may depend on M0."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-efficient-features_amd", "csrc")
FLAGS = "-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -S".split()
UNITS = ["detect_kernels.hip", "bad_kernel.hip", "hashsift_kernels.hip", "match_kernels.hip", "input_kernels.hip"]


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc (ROCm) on this machine")
    out = {}
    d = tmp_path_factory.mktemp("isa")
    for u in UNITS:
        path = str(d / (u + ".s"))
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, u), "-o", path], cwd=CSRC, stderr=subprocess.DEVNULL)
        out[u] = open(path).read()
    return out


def _kernels(text):
    """name -> assembly text of every kernel of a translation unit"""
    res = {}
    for m in re.finditer(r"^(_Z\w+):.*?\.end_amdhsa_kernel", text, flags=re.S | re.M):
        res[m.group(1)] = m.group(0)
    return res


def test_no_kernel_uses_scratch(asm):
    bad = []
    for u, text in asm.items():
        ks = _kernels(text)
        assert ks, u
        for name, body in ks.items():
            m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
            assert m, name
            if int(m.group(1)) != 0:
                bad.append((u, name, int(m.group(1))))
    assert not bad, f"kernels with a private segment (spills): {bad}"


def test_rows_kernel_owns_m0(asm):
    ks = {n: b for n, b in _kernels(asm["detect_kernels.hip"]).items() if "resize_rows_kernel" in n}
    assert len(ks) == 4, list(ks)                                   # NLEV = 1 .. 4
    for name, body in ks.items():
        outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body, flags=re.S)
        uses = [l.strip() for l in outside.splitlines() if re.search(r"\bm0\b", l) and not l.strip().startswith((";", "."))]
        assert not uses, f"{name}: the compiler uses M0 outside the LDS-DMA statements: {uses[:4]}"
        assert body.count("offen lds") >= 4 and "s_waitcnt vmcnt(10)" in body and "s_waitcnt vmcnt(8)" in body
