"""The device's cosf / sinf for HashSIFT's patch rotation (csrc/glibc_sincosf.h, a restatement of glibc's sincosf) against
the host libm the reference CPU code would call (hash_sift.cpp:121-122): every float in [2^-13, 11), bit for bit."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cmath>
#include <cstdio>
#include <cstring>
#include "glibc_sincosf.h"
int main() {
    long bc = 0, bs = 0, tot = 0;
    float a = 0x1p-13f, lim = EFX_GLIBC_SINCOSF_MAX;
    uint32_t u0, u1; memcpy(&u0, &a, 4); memcpy(&u1, &lim, 4);
    for (uint32_t u = u0; u < u1; u++) {
        float x; memcpy(&x, &u, 4);
        if (cosf(x) != efx_glibc_sincosf(x, 1)) bc++;
        if (sinf(x) != efx_glibc_sincosf(x, 0)) bs++;
        tot++;
    }
    // below 2^-13: cosf = 1, sinf = x
    if (efx_glibc_sincosf(0.f, 1) != 1.f || efx_glibc_sincosf(0.f, 0) != 0.f || efx_glibc_sincosf(1e-5f, 0) != sinf(1e-5f)) bc++;
    printf("%ld %ld %ld\n", tot, bc, bs);
    return 0;
}
"""


def test_sincosf_restatement_equals_host_libm_exhaustively():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "cuda-efficient-features_amd", "csrc"),
                        "-o", exe, src], check=True)
        tot, bc, bs = map(int, subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert tot > 130_000_000
    assert bc == 0 and bs == 0, f"{bc} cosf and {bs} sinf values differ from the host libm"
