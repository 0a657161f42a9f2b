"""the HPatches exporter (samples/hpatches_description.cpp): CSV bit format (:76-105), calcUMax
(:107-126), ICAngles (:128-162), fixed keypoints of size 64 / angle -1 (:231-249)."""
import math
import os

import numpy as np
import pytest

from tools import png8, synth


def test_umax_and_fast_atan2(oracle):
    # calcUMax(31) is ORB's table, the one cuda_efficient_features.cu hard-codes
    assert oracle.calc_umax(31) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3, 0]
    u = oracle.calc_umax(65)
    assert len(u) == 34 and u[0] == 32 and u[33] == 0
    # circle symmetry: row v ends at umax[v], and umax[umax[v]] >= v for the rows above the diagonal
    assert all(u[v] >= u[v + 1] for v in range(32))
    rng = np.random.default_rng(0)
    for y, x in rng.integers(-100000, 100000, size=(300, 2)):
        want = math.degrees(math.atan2(float(y), float(x))) % 360.0
        got = float(oracle.fast_atan2(float(y), float(x)))
        assert abs((got - want + 180) % 360 - 180) < 0.3          # cv::fastAtan2's documented accuracy
    assert float(oracle.fast_atan2(0.0, 0.0)) == 0.0 and float(oracle.fast_atan2(0.0, -5.0)) == 180.0


def test_ic_angles_match_numpy_moments(oracle):
    img = synth.synth_frame(260, 325, seed=5, density=2.0)
    kp4 = np.array([[32.5, 32.5, 64, -1], [97.5, 162.5, 64, -1], [292.5, 227.5, 64, -1], [3.2, 5.9, 64, -1]], np.float32)
    got = oracle.ic_angles(img, kp4, 65)
    umax = oracle.calc_umax(65)
    I = np.zeros((260 + 80, 325 + 80), np.int64)
    I[40:300, 40:365] = img
    for i in range(len(kp4)):
        cx, cy = int(math.floor(kp4[i, 0])) + 40, int(math.floor(kp4[i, 1])) + 40
        m01 = m10 = 0
        for v in range(-32, 33):
            d = umax[abs(v)] if v != 0 else 32
            for uu in range(-d, d + 1):
                m10 += uu * I[cy + v, cx + uu]
                m01 += v * I[cy + v, cx + uu]
        assert got[i, 3] == oracle.fast_atan2(float(m01), float(m10))
    assert np.array_equal(got[:, :3], kp4[:, :3])


def test_png_codec_round_trip(tmp_path):
    img = synth.synth_frame(130, 65, seed=2)
    p = str(tmp_path / "a.png")
    png8.write(p, img)
    assert np.array_equal(png8.read(p), img)


def make_hpatches_tree(root, nseq=2, npatches=3, nimages=3):
    names = ["ref", "e1", "h2"]
    for s in range(nseq):
        d = os.path.join(root, "v_seq%d" % s)
        os.makedirs(d)
        for j in range(nimages):
            png8.write(os.path.join(d, names[j] + ".png"), synth.synth_frame(npatches * 65, 65, seed=100 * s + j, density=3.0))
    return names[:nimages]


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def cef():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cef_loader
    return cef_loader.load()


@pytest.mark.gpu
def test_csv_format(cef):
    d = np.array([[0b10110000, 0x01], [0xFF, 0x00]], np.uint8)
    assert cef.descriptorsToCsv(d) == "1,0,1,1,0,0,0,0,0,0,0,0,0,0,0,1\n1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0\n"


@pytest.mark.gpu
def test_device_ic_angles_equal_oracle(cef, oracle):
    img = synth.synth_frame(390, 455, seed=7, density=2.0)
    kps = cef.keypoints_array(42)
    k = 0
    for x in range(7):
        for y in range(6):
            kps[k]["x"], kps[k]["y"], kps[k]["size"], kps[k]["angle"] = 65 * (x + 0.5), 65 * (y + 0.5), 64, -1
            k += 1
    got = cef.icAngles(img, kps, 65)
    kp4 = np.stack([kps["x"], kps["y"], kps["size"], kps["angle"]], 1)
    want = oracle.ic_angles(img, kp4, 65)
    assert np.array_equal(got["angle"], want[:, 3])
    assert np.array_equal(cef.icAngles(img, kps, 31)["angle"], oracle.ic_angles(img, kp4, 31)[:, 3])


@pytest.mark.gpu
@pytest.mark.parametrize("desc_type,bits,angle", [(0, 256, False), (0, 512, True), (1, 256, False)])
def test_exporter_end_to_end(cef, oracle, tmp_path, desc_type, bits, angle):
    """The exporter's CSV files hold exactly the oracle's descriptor bits for size-64 keypoints at the patch centres."""
    from tools import hpatches_description as H
    root = str(tmp_path / "hp")
    names = make_hpatches_tree(root)
    out = str(tmp_path / "res")
    H.main([root, "--result-dir", out, "--descriptor-type", str(desc_type), "--descriptor-bits", str(bits)] + (["--compute-angle"] if angle else []))
    for s in range(2):
        seq = "v_seq%d" % s
        imgs = [png8.read(os.path.join(root, seq, n + ".png")) for n in sorted(names)]
        stacked = np.ascontiguousarray(np.concatenate(imgs, axis=1))
        kp4 = np.array([[65 * (x + 0.5), 65 * (y + 0.5), 64, -1] for x in range(3) for y in range(3)], np.float32)
        if angle:
            kp4 = oracle.ic_angles(stacked, kp4, 65)
        want = oracle.bad_compute(stacked, kp4, bits) if desc_type == 0 else oracle.hashsift_compute(stacked, kp4, bits)
        for x, n in enumerate(sorted(names)):
            text = open(os.path.join(out, "%s_%d" % (H.DESC_STR[desc_type], bits), seq, n + ".csv")).read()
            rows = [np.packbits(np.array(line.split(","), dtype=np.uint8)) for line in text.strip().split("\n")]
            got = np.stack(rows)
            if desc_type == 0:
                assert np.array_equal(got, want[3 * x:3 * x + 3])
            else:
                assert np.count_nonzero(got != want[3 * x:3 * x + 3]) <= 1
