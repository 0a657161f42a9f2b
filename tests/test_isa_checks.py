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


def test_select_kernel_waits_for_its_stores_before_the_hand_offs(asm):
    """select_kernel hands results from workgroup to workgroup inside one launch with relaxed device-scope stores followed by an
    atomic counter / a flag word.  The stores must have been ACKNOWLEDGED before the counter or the flag goes out; a workgroup-scope
    release fence does not do that on gfx950 (it waits for LDS traffic only) -- the race of docs/history/round6.md section 11, one 8K
    frame in ~500 000.  The waits are explicit `s_waitcnt vmcnt(0)` statements; this test keeps them where they belong:
      * behind the counting workgroups' count stores (`global_store_dword ... sc1`) and before the barrier that precedes the
        `done` counter;
      * between the leader's two published words (two `global_store_dwordx2 ... sc1` of neighbouring offsets)."""
    ks = {n: b for n, b in _kernels(asm["detect_kernels.hip"]).items() if "select_kernel" in n}
    assert len(ks) == 1, list(ks)
    body = next(iter(ks.values()))
    lines = body.splitlines()
    explicit = [i for i, l in enumerate(lines) if l.strip() == "s_waitcnt vmcnt(0)" and i > 0 and "ASMSTART" in lines[i - 1]]
    assert len(explicit) >= 2, "the explicit waits for the stores' acknowledgement are gone"
    # the count store, then the explicit wait, then the barrier -- in this order, nothing else that stores in between
    count_store = [i for i, l in enumerate(lines) if re.search(r"global_store_dword v\[\d+:\d+\], v\d+, off sc1", l)]
    assert count_store, "the counting pass's device-scope count store was not found"
    i0 = count_store[-1]
    barrier = next(i for i in range(i0, len(lines)) if lines[i].strip() == "s_barrier")
    assert any(i0 < i < barrier for i in explicit), "no wait for the count stores before the barrier of the done counter"
    # the leader: word B (higher offset), the wait, word A
    pub = [(i, int(re.search(r"offset:(\d+) sc1", l).group(1))) for i, l in enumerate(lines)
           if re.search(r"global_store_dwordx2 v\d+, v\[\d+:\d+\], s\[\d+:\d+\] offset:\d+ sc1", l)]
    pairs = [(a, b) for a, b in zip(pub, pub[1:]) if a[1] == b[1] + 8]
    assert pairs, f"the leader's two published words were not found: {pub}"
    (ib, _), (ia, _) = pairs[-1]
    assert any(ib < i < ia for i in explicit), "no wait between the leader's word B and word A"
