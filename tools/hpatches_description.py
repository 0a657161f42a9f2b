#!/usr/bin/env python3
"""HPatches descriptor exporter -- the reference's samples/hpatches_description.cpp on the HIP library.

    python tools/hpatches_description.py <hpatches-dir> [--result-dir DIR] [--descriptor-type 0|1] [--descriptor-bits 256|512]
                                         [--compute-angle]

For every sequence directory: the PNG patch strips (65-px wide columns of 65x65 patches) are loaded as gray images and
concatenated horizontally (:216-221); one keypoint per patch at the patch centre, size 64, angle -1 (:231-241), or the
intensity-centroid angle of the 65-px circular patch with --compute-angle (:243-244, ICAngles :128-162); descriptors
with EfficientFeatures::compute on the stacked RAW image (:247-248); one CSV per input image, one line per patch, bits
MSB first (:76-105, :250-258)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tools import png8

PATCH_SIZE = 65
DESC_STR = ["BAD", "HashSIFT"]


def descriptor_enum(EF, desc_type, bits):
    """getDescriptorType of samples/sample_common.cpp:24-33."""
    if desc_type == 0:
        return EF.BAD_256 if bits == 256 else EF.BAD_512
    if desc_type == 1:
        return EF.HASH_SIFT_256 if bits == 256 else EF.HASH_SIFT_512
    return EF.HASH_SIFT_256


def patch_keypoints(cef, npatches, nimages):
    kps = cef.keypoints_array(npatches * nimages)
    i = 0
    for x in range(nimages):                                   # hpatches_description.cpp:234-241
        for y in range(npatches):
            kps[i]["x"] = np.float32(PATCH_SIZE) * np.float32(x + 0.5)
            kps[i]["y"] = np.float32(PATCH_SIZE) * np.float32(y + 0.5)
            kps[i]["size"] = 64.0
            kps[i]["angle"] = -1.0
            kps[i]["response"] = 0.0
            kps[i]["octave"] = 0
            kps[i]["class_id"] = -1
            i += 1
    return kps


def export_sequence(cef, feature, seq_dir, save_dir, compute_angle):
    files = sorted(f for f in os.listdir(seq_dir) if f.lower().endswith(".png"))
    if not files:
        return 0
    images = [png8.read(os.path.join(seq_dir, f)) for f in files]
    stacked = np.ascontiguousarray(np.concatenate(images, axis=1))        # cv::hconcat
    npatches, nimages = stacked.shape[0] // PATCH_SIZE, len(images)
    kps = patch_keypoints(cef, npatches, nimages)
    if compute_angle:
        kps = cef.icAngles(stacked, kps, PATCH_SIZE)
    desc = feature.compute(stacked, kps)
    os.makedirs(save_dir, exist_ok=True)
    for x, f in enumerate(files):
        with open(os.path.join(save_dir, os.path.splitext(f)[0] + ".csv"), "w") as out:
            out.write(cef.descriptorsToCsv(desc[x * npatches:(x + 1) * npatches]))
    return npatches * nimages


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("hpatches_dir")
    ap.add_argument("--result-dir", default="./results")
    ap.add_argument("--descriptor-type", type=int, default=0, help="0: BAD 1: HashSIFT")
    ap.add_argument("--descriptor-bits", type=int, default=256)
    ap.add_argument("--compute-angle", action="store_true")
    a = ap.parse_args(argv)
    import cef_loader
    cef = cef_loader.load()
    EF = cef.EfficientFeatures
    feature = EF.create()
    feature.setDescriptorType(descriptor_enum(EF, a.descriptor_type, a.descriptor_bits))
    desc_dir = os.path.join(a.result_dir, "%s_%d" % (DESC_STR[a.descriptor_type], a.descriptor_bits))
    seqs = sorted(d for d in os.listdir(a.hpatches_dir) if os.path.isdir(os.path.join(a.hpatches_dir, d)))
    print("number of patch directories:", len(seqs))
    for i, s in enumerate(seqs):
        n = export_sequence(cef, feature, os.path.join(a.hpatches_dir, s), os.path.join(desc_dir, s), a.compute_angle)
        print("sequence: %3d/%3d [%s] %d patches" % (i + 1, len(seqs), s, n))


if __name__ == "__main__":
    main()
