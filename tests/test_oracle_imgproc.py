"""Oracle image stages (oracle/imgproc_oracle.c) against the committed scipy/numpy goldens and against known answers
derived from the reference code.  CPU only."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from miunet import synth


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "imgproc.npz"))


# ---------------------------------------------------------------- A2/A3 preprocess_raw (src/preprocess.cpp:65-118)
@pytest.mark.parametrize("j", [0, 1, 2, 3])
def test_preprocess_matches_numpy_golden(gold, j):
    h, w, seed = (int(v) for v in gold[f"raw_shape{j}"])
    raw = synth.make_raw16(h, w, seed=seed)
    assert np.array_equal(orc.preprocess_raw(raw), gold[f"pre{j}"])


def test_preprocess_known_answers():
    const = np.full((40, 30), 1234, np.uint16)                      # mn == mx -> mx = mn + 1 -> all zeros
    assert not orc.preprocess_raw(const).any()
    wrap = np.full((8, 8), 65535, np.uint16)                        # mn == mx == 65535: mx wraps to 0, scale negative
    out = orc.preprocess_raw(wrap)                                  # (v - mn) * scale + 0.5 = 0.5 -> 0
    assert not out.any()
    ramp = np.array([[0, 1000], [2000, 3000]], np.uint16)           # 2x2 -> 512x512 up-sampling, clamped taps
    out = orc.preprocess_raw(ramp)
    assert out[0, 0] == 0 and out[511, 511] == 255                  # exact 255 at the maximum (last taps clamp to it)
    assert out[0, 256] == int(1000 / 3000 * 255 + 0.5) and out[256, 0] == int(2000 / 3000 * 255 + 0.5)
    assert np.all(np.diff(out[0].astype(int)) >= 0) and np.all(np.diff(out[:, 0].astype(int)) >= 0)
    # top-left alignment: output x samples input x * (w/512) exactly when w is a multiple of 512
    big = (np.arange(1024, dtype=np.uint16)[None, :] * 60 + np.zeros((1024, 1), np.uint16)).astype(np.uint16)
    out = orc.preprocess_raw(big)
    want = ((big[0, ::2].astype(np.float64) - 0) * (255.0 / big.max()) + 0.5).astype(np.uint8)
    assert np.array_equal(out[0], want)


# ---------------------------------------------------------------- A10 LUT (src/process.cpp:178-185)
def test_mask_to_image_lut():
    m = np.arange(256, dtype=np.uint8).reshape(16, 16)
    v = orc.mask_to_image(m)
    want = np.zeros(256, np.uint8); want[1] = 128; want[2] = 255
    assert np.array_equal(v.reshape(-1), want)


# ---------------------------------------------------------------- A8/A9 postprocess (src/postprocess.cpp:13-79)
@pytest.mark.parametrize("i", [0, 1, 2, 3, 4])
def test_postprocess_matches_scipy_golden(gold, i):
    m = gold[f"mask{i}"]
    assert np.array_equal(orc.fill_holes(m), gold[f"filled{i}"])
    assert np.array_equal(orc.open3x3((gold[f"filled{i}"] == 2).astype(np.uint8) * 255), gold[f"opened{i}"])
    out = orc.postprocess_mask(m)
    assert np.array_equal(out, gold[f"final{i}"])
    assert set(np.unique(out)) <= {0, 2}


def test_hole_threshold_is_15728_at_512():
    # min_area = int(w * h * 0.06f) evaluated in float (src/postprocess.cpp:30); a hole is filled iff area < min_area (:40)
    assert int(np.float32(512 * 512) * np.float32(0.06)) == 15728
    base = np.zeros((512, 512), np.uint8)
    base[4:508, 4:508] = 2
    ex = base.copy()
    ex[100:132, 10:501] = 0
    ex[132, 10:26] = 0                                               # 32 * 491 + 16 = 15728 -> NOT filled
    assert (orc.fill_holes(ex)[100:132, 10:501] == 0).all()
    ex[132, 25] = 2                                                  # 15727 -> filled
    out = orc.fill_holes(ex)
    assert (out[100:133, 10:501] == 2).all()


def test_hole_touching_edge_or_leaking_diagonally_is_not_filled():
    m = np.zeros((64, 64), np.uint8); m[:, :] = 2
    m[10:20, 0:5] = 0                                                # touches x = 0
    m[30:35, 59:64] = 1                                              # class 1 touching x = w-1
    out = orc.fill_holes(m)
    assert (out[10:20, 0:5] == 0).all() and (out[30:35, 59:64] == 1).all()
    m = np.zeros((64, 64), np.uint8); m[5:60, 5:60] = 2
    m[20:25, 20:25] = 0
    m[25, 25] = 0; m[26, 26] = 0                                     # diagonal chain (8-connectivity) ...
    for k in range(27, 61):
        m[k, k] = 0                                                  # ... out to the background
    out = orc.fill_holes(m)
    assert (out[20:25, 20:25] == 0).all()
    m2 = np.zeros((64, 64), np.uint8); m2[5:60, 5:60] = 2; m2[20:25, 20:25] = 1
    assert (orc.fill_holes(m2)[20:25, 20:25] == 2).all()             # class-1 hole is filled to 2


def test_area_filter_threshold_and_alphabet():
    # a component is kept iff area >= min_area = 15728 (src/postprocess.cpp:70); 15728 = 32 * 491 + 16 (983 is prime, so
    # no rectangle has that area): a 32 x 491 block with a 16-px-wide one-row step below it is invariant under the 3x3 open
    m = np.zeros((512, 512), np.uint8)
    m[10:42, 10:501] = 2
    m[42, 10:26] = 2
    out = orc.postprocess_mask(m)
    assert (out == 2).sum() == 15728 and np.array_equal(out, m)
    m[42, 25] = 0                                                    # 15727 -> dropped
    assert not orc.postprocess_mask(m).any()
    m = np.zeros((512, 512), np.uint8); m[10:42, 10:501] = 2; m[42, 10] = 2; m[43:60, 10] = 2   # 1-px-wide tail: removed by the open
    out = orc.postprocess_mask(m)
    assert not out.any()                                             # 15712 < 15728 after the tail is gone
    m = np.zeros((512, 512), np.uint8); m[100:300, 100:300] = 2; m[5, 5] = 2; m[300:302, 200:202] = 2; m[302:420, 150:400] = 2
    m[150:160, 150:160] = 1                                          # class-1 island inside: a hole -> filled
    out = orc.postprocess_mask(m)
    assert out[5, 5] == 0 and out[300, 200] == 0 and out[150, 150] == 2 and out[350, 200] == 2
    assert set(np.unique(out)) <= {0, 2}


def test_connected_components_stats():
    fg = np.zeros((8, 10), np.uint8)
    fg[1:3, 1:4] = 255; fg[3, 4] = 255                               # diagonal touch joins (8-connectivity)
    fg[6, 0] = 255
    nc, labels, stats = orc.connected_components8(fg)
    assert nc == 3
    areas = sorted(stats[1:, 4].tolist())
    assert areas == [1, 7]
    big = stats[1:][np.argmax(stats[1:, 4])]
    assert big.tolist() == [1, 1, 4, 3, 7]                           # left, top, width, height, area


# ---------------------------------------------------------------- A11/A12 contours (src/mask2polygon.cpp:29-63)
def test_contours_known_answers():
    m = np.zeros((12, 16), np.uint8); m[2:6, 3:9] = 255
    assert orc.find_contours(m) == [[(3, 2), (3, 5), (8, 5), (8, 2)]]            # TL, BL, BR, TR
    m[8, 1] = 255; m[10, 5:9] = 255
    assert orc.find_contours(m) == [[(5, 10), (8, 10)], [(1, 8)], [(3, 2), (3, 5), (8, 5), (8, 2)]]   # newest first
    m = np.zeros((10, 10), np.uint8); m[1:9, 1:9] = 255; m[3:7, 3:7] = 0; m[4:6, 4:6] = 255
    assert orc.find_contours(m) == [[(1, 1), (1, 8), (8, 8), (8, 1)]]            # nested blob in a hole is not external
    m = np.zeros((6, 6), np.uint8); m[:, :] = 255
    assert orc.find_contours(m) == [[(0, 0), (0, 5), (5, 5), (5, 0)]]            # blob touching every edge
    m = np.zeros((5, 5), np.uint8); m[1, 1] = 200; m[2, 2] = 128; m[3, 3] = 127   # threshold: > 127 only
    assert orc.find_contours(m) == [[(1, 1), (2, 2)]]
    assert orc.find_contours(np.zeros((4, 4), np.uint8)) == []
    # a diamond: every step changes direction only at the 4 corners
    m = np.zeros((9, 9), np.uint8)
    for r in range(9):
        for c in range(9):
            if abs(r - 4) + abs(c - 4) <= 3:
                m[r, c] = 255
    assert orc.find_contours(m) == [[(4, 1), (1, 4), (4, 7), (7, 4)]]


def test_contour_point_mapping_truncates():
    assert orc.map_points([(511, 511), (3, 7), (0, 0)], 2048 / 512.0, 1536 / 512.0) == [(2044, 1533), (12, 21), (0, 0)]
    assert orc.map_points([(5, 5)], 300 / 512.0, 200 / 512.0) == [(2, 1)]         # 2.93 -> 2, 1.95 -> 1
