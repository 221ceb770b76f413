"""postprocess_mask on the device (union-find labelling, 3x3 open, area filter), integer-exact
against the oracle's restatement of src/postprocess.cpp:13-79 and the scipy goldens."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights

pytestmark = pytest.mark.gpu


def _blobs(seed, h, w, fill):
    rng = np.random.default_rng(seed)
    f = rng.random((h, w))
    for _ in range(4):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5
    m = np.where(f > np.quantile(f, 1 - fill), 2, 0).astype(np.uint8)
    m[rng.random((h, w)) < 0.03] = 1
    m[rng.random((h, w)) < 0.02] = 2
    return m


def test_goldens_and_known_answers_512(golden_dir):
    g = np.load(os.path.join(golden_dir, "imgproc.npz"))
    masks = [g[f"mask{i}"] for i in range(4)]                                    # the four 512x512 goldens
    ex = np.zeros((512, 512), np.uint8); ex[4:508, 4:508] = 2
    ex[100:132, 10:501] = 0; ex[132, 10:26] = 0                                  # hole of exactly 15728 px: NOT filled
    ex2 = ex.copy(); ex2[132, 25] = 2                                            # 15727 px: filled
    keep = np.zeros((512, 512), np.uint8); keep[10:42, 10:501] = 2; keep[42, 10:26] = 2    # component of exactly 15728: kept
    drop = keep.copy(); drop[42, 25] = 0                                         # 15727: dropped
    spk = (np.random.default_rng(5).integers(0, 3, (512, 512))).astype(np.uint8)  # speckle: heavy union-find contention
    full = np.full((512, 512), 2, np.uint8)
    batch = np.stack(masks + [ex, ex2, keep, drop, spk, full, np.zeros((512, 512), np.uint8)] + [_blobs(s, 512, 512, 0.5) for s in range(7)])
    with binding.Engine(512, 512, max_batch=8) as eng:                            # 18 masks through micro-batches of 8, 8, 2
        got = eng.postprocess_masks(batch)
    for i in range(4):
        assert np.array_equal(got[i], g[f"final{i}"])
    for i in range(batch.shape[0]):
        assert np.array_equal(got[i], orc.postprocess_mask(batch[i])), i
    assert set(np.unique(got)) <= {0, 2}


@pytest.mark.parametrize("h,w", [(64, 64), (96, 160), (48, 80)])
def test_random_masks_small(h, w):
    batch = np.stack([_blobs(100 + s, h, w, f) for s, f in enumerate([0.2, 0.35, 0.5, 0.65, 0.8, 0.5])])
    with binding.Engine(h, w, max_batch=4) as eng:
        got = eng.postprocess_masks(batch)
    for i in range(batch.shape[0]):
        assert np.array_equal(got[i], orc.postprocess_mask(batch[i])), i


def test_fused_into_inference():
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_threshold_weights(spec))
    raws = [synth.make_raw16(1536, 2048, seed=21), synth.make_raw16(600, 800, seed=41)]
    with binding.Engine(512, 512, max_batch=2) as eng:
        eng.load_weights(blob)
        tiles, labels, _ = eng.infer_raw16(raws)
        eng.set_postprocess(True)
        tiles2, post, _ = eng.infer_raw16(raws)
        post_u8, _ = eng.infer(tiles[..., None])
    assert np.array_equal(tiles, tiles2)
    for i in range(2):
        assert np.array_equal(post[i], orc.postprocess_mask(labels[i]))
        assert np.array_equal(post_u8[i], post[i])
    assert (post == 2).any()
