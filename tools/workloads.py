"""The BASELINE.json configurations as seeded synthetic workloads, shared by tests/test_gpu_fullsize.py,
tools/bench_configs.py and bench.py (so that what is tested is what is timed).

  C2  4K frame (seed 1000), detect-only, reference defaults, nfeatures 40 000
  C3  4K frame, compute-only BAD256 / BAD512 on 40 000 keypoints
  C4  4K frame, compute-only HashSIFT512 on 40 000 keypoints
  C5  8K frames (seeds 1000 + k), detectAndCompute BAD512, nfeatures 40 000

"40k keypoints on a 4K frame": with the reference's default NMS radius (15 px) a 4K pyramid saturates at ~22 000
survivors whatever the content (level 7 would need 73 % of the densest possible packing), so the detector's defaults
cannot produce the stated workload.  C3 / C4 therefore take their keypoints from the detector run with NMS radius 5 on a
denser frame (density 0.6): every level fills its quota and N is EXACTLY 40 000 (8687 + 7239 + 6033 + 5027 + 4189 + 3491
+ 2909 + 2425), spread over all eight levels as the reference's benchmark keypoints are (sample_benchmark.cpp:132-141:
compute() on the keypoints detect() returned).
"""
from tools import synth

K4 = (2160, 3840)
K8 = (4320, 7680)
C34_DENSITY = 0.6
C34_NMS_RADIUS = 5
N40K = 40000


def frame_c2(seed=1000):
    return synth.synth_frame(K4[0], K4[1], seed=seed)


def frame_c34(seed=1000):
    """4K frame for the compute-only configurations; detect with nonmax_radius=C34_NMS_RADIUS, nfeatures=N40K."""
    return synth.synth_frame(K4[0], K4[1], seed=seed, density=C34_DENSITY)


def frame_c5(k=0):
    return synth.synth_frame(K8[0], K8[1], seed=1000 + k)
