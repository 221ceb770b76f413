"""Oracle (oracle/unet_oracle.c) against the committed PyTorch-CPU golden vectors and against known answers
derived from the reference's code (src/process.cpp:22-42, :158-170).  CPU only."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as orc
from miunet import synth
from miunet.spec import UNetSpec, pack_weights, unpack_weights


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "unet_*.npz")))


def load_case(path):
    z = np.load(path)
    in_ch, base, levels, classes, b, h, w, wseed, iseed = (int(v) for v in z["meta"])
    spec = UNetSpec(in_ch, base, levels, classes)
    blob = pack_weights(spec, synth.make_weights(spec, wseed))
    imgs = synth.make_images(b, h, w, in_ch, iseed, str(z["kind"]))
    return spec, blob, imgs, z["logits"]


def test_golden_files_exist(golden_dir):
    assert len(_cases(golden_dir)) >= 4


@pytest.mark.parametrize("name", ["unet_b64_l4_64", "unet_b64_l4_48x80", "unet_b16_l3_40x24", "unet_b32_l5_c3_64"])
def test_oracle_matches_torch_golden(golden_dir, name):
    spec, blob, imgs, want = load_case(os.path.join(golden_dir, name + ".npz"))
    logits, labels = orc.unet_forward(blob, imgs)
    # two independent fp32 implementations with different summation orders: 1e-4 absolute on O(1) logits
    assert np.max(np.abs(logits - want)) < 1e-4
    # labels follow the oracle's own logits exactly (first max wins)
    assert np.array_equal(labels, np.argmax(logits, axis=1).astype(np.uint8))
    # and agree with the golden argmax wherever the golden top-2 margin exceeds the logit tolerance
    srt = np.sort(want, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert np.array_equal(labels[safe], np.argmax(want, axis=1).astype(np.uint8)[safe])


def test_weight_file_roundtrip():
    spec = UNetSpec(1, 16, 2, 3)
    t = synth.make_weights(spec, 3)
    blob = pack_weights(spec, t)
    spec2, t2 = unpack_weights(blob)
    assert (spec2.in_ch, spec2.base, spec2.levels, spec2.classes) == (spec.in_ch, spec.base, spec.levels, spec.classes)
    assert abs(spec2.bn_eps - spec.bn_eps) < 1e-12
    for k in t:
        assert np.array_equal(t[k], t2[k])
    assert spec.n_params() * 4 + 36 == len(blob)


def test_param_and_mac_count_match_survey():
    spec = UNetSpec()
    learnable = sum(int(np.prod(s)) for n, s in spec.tensor_list() if not n.endswith((".mean", ".var")))
    assert learnable == 31036611
    # 192 401 113 088 MAC / image at 512x512
    assert spec.macs_per_image(512, 512) == 192401113088


def test_normalize_is_true_division():
    # A4, src/process.cpp:36-39: all 256 codes equal i/255.0f bit for bit (and differ from i*(1/255.0f) for many)
    codes = np.arange(256, dtype=np.uint8)
    got = orc.normalize_u8(codes)
    want = codes.astype(np.float32) / np.float32(255.0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    recip = codes.astype(np.float32) * (np.float32(1.0) / np.float32(255.0))
    assert np.count_nonzero(recip.view(np.uint32) != want.view(np.uint32)) > 50


def test_argmax_rules():
    # A7, src/process.cpp:158-170
    fmax = np.finfo(np.float32).max
    px = np.array([
        [1.0, 1.0, 1.0],            # tie -> lowest index
        [0.0, 2.0, 2.0],            # tie between 1 and 2 -> 1
        [np.nan, np.nan, np.nan],   # NaN never selected -> 0
        [np.nan, -1.0, np.nan],     # -> 1
        [-fmax, -fmax, -fmax],      # nothing is > -FLT_MAX -> 0
        [-np.inf, -np.inf, -np.inf],
        [-fmax, -fmax, -1e38],      # -> 2
        [3.0, -1.0, 2.0],
        [np.nan, 5.0, 7.0],
    ], dtype=np.float32)
    logits = np.ascontiguousarray(px.T.reshape(3, 1, -1))
    got = orc.argmax_planar(logits)[0]
    assert got.tolist() == [0, 1, 0, 1, 0, 0, 2, 0, 2]


def test_batch_independence():
    spec = UNetSpec(1, 16, 2, 3)
    blob = pack_weights(spec, synth.make_weights(spec, 11))
    imgs = synth.make_images(3, 16, 24, 1, 5)
    lg, lb = orc.unet_forward(blob, imgs)
    lg1, lb1 = orc.unet_forward(blob, imgs[1:2])
    assert np.array_equal(lg[1], lg1[0]) and np.array_equal(lb[1], lb1[0])


def test_thread_count_does_not_change_bits():
    spec = UNetSpec(1, 16, 2, 3)
    blob = pack_weights(spec, synth.make_weights(spec, 11))
    imgs = synth.make_images(1, 32, 32, 1, 9)
    a, _ = orc.unet_forward(blob, imgs, nthreads=1)
    b, _ = orc.unet_forward(blob, imgs, nthreads=4)
    assert np.array_equal(a, b)
