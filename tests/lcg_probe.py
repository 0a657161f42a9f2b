"""The SURVEY.md Appendix B probe input: LCG-noise 640x480 image + 200 keypoints.

s = s*1664525 + 1013904223 (mod 2^32), seed 12345.  Pixel = s >> 24 (one draw per pixel, raster
order); then per keypoint three draws: x = (s>>8) % W, y = (s>>8) % H, angle = ((s>>8) % 36000)/100;
size 31.  The draw order was recovered by matching the recorded reference hashes."""
import numpy as np

W, H, N = 640, 480, 200


def lcg_stream(seed, n):
    out = np.empty(n, dtype=np.uint64)
    s = seed
    for i in range(n):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = s
    return out


def probe_input():
    st = lcg_stream(12345, W * H + 3 * N)
    img = ((st[:W * H] >> 24) & 0xFF).astype(np.uint8).reshape(H, W)
    v = st[W * H:] >> 8
    k = np.zeros((N, 4), np.float32)
    k[:, 0] = (v[0::3] % W)
    k[:, 1] = (v[1::3] % H)
    k[:, 2] = 31
    k[:, 3] = (v[2::3] % 36000).astype(np.float64) / 100
    return img, k


def fnv1a32(arr):
    h = 0x811C9DC5
    for x in np.ascontiguousarray(arr).tobytes():
        h ^= x
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


# FNV-1a-32 of the descriptor bytes produced by the REFERENCE CPU code
REFERENCE_HASHES = {("bad", 256): 0x395D24F6, ("bad", 512): 0xDE027879,
                    ("hashsift", 256): 0xE3FF078E, ("hashsift", 512): 0x947263FC}
