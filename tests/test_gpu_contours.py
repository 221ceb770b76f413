"""extract_contours (threshold 127 + external contours, CHAIN_APPROX_SIMPLE, newest first) on the
device, exact against the oracle's sequential Suzuki-Abe restatement (oracle/imgproc_oracle.c)."""
import numpy as np
import pytest

import oracle_lib as orc
from miunet import binding

pytestmark = pytest.mark.gpu


def _blob_mask(seed, h, w, smooth=2, q=0.55):
    rng = np.random.default_rng(seed)
    f = rng.random((h, w))
    for _ in range(smooth):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5
    m = (f > np.quantile(f, q)).astype(np.uint8) * 255
    m[rng.random((h, w)) < 0.02] = 255
    m[rng.random((h, w)) < 0.02] = 0
    return m


def test_known_answers_and_nesting():
    masks = np.zeros((7, 64, 64), np.uint8)
    masks[0, 2:6, 3:9] = 255                                            # rectangle: TL, BL, BR, TR
    masks[1, 2:6, 3:9] = 255; masks[1, 8, 1] = 200; masks[1, 10, 5:9] = 128   # + isolated pixel + 1-px line, newest first
    masks[2, 1:30, 1:30] = 255; masks[2, 5:25, 5:25] = 0; masks[2, 8:20, 8:20] = 255; masks[2, 11:15, 11:15] = 0; masks[2, 12:14, 12:14] = 255
    masks[3, :, :] = 255                                                 # touches every edge
    masks[4, 0, 0] = 255; masks[4, 63, 63] = 255; masks[4, 0:3, 63] = 255
    masks[5, 10:40, 10:40] = 255; masks[5, 20:30, 20:40] = 0; masks[5, 22:28, 25:35] = 255    # C-shape open to the right + blob in the notch
    masks[6, 30, 30] = 127; masks[6, 31, 31] = 128                      # threshold is strict (> 127)
    with binding.Engine(64, 64, max_batch=4) as eng:
        got = eng.extract_contours(masks, cap_points=2048, cap_contours=64)
    assert got[0] == [[(3, 2), (3, 5), (8, 5), (8, 2)]]
    assert got[1] == [[(5, 10), (8, 10)], [(1, 8)], [(3, 2), (3, 5), (8, 5), (8, 2)]]
    assert got[2] == [[(1, 1), (1, 29), (29, 29), (29, 1)]]             # everything nested inside the hole is not external
    assert got[3] == [[(0, 0), (0, 63), (63, 63), (63, 0)]]
    assert got[6] == [[(31, 31)]]
    for i in range(7):
        assert got[i] == orc.find_contours(masks[i]), i


@pytest.mark.parametrize("h,w,smooth,q", [(64, 64, 2, 0.55), (96, 160, 1, 0.5), (48, 80, 3, 0.6), (128, 128, 0, 0.5)])
def test_random_masks_vs_oracle(h, w, smooth, q):
    masks = np.stack([_blob_mask(1000 * smooth + s, h, w, smooth, q) for s in range(6)])
    with binding.Engine(h, w, max_batch=4) as eng:
        got = eng.extract_contours(masks, cap_points=h * w, cap_contours=h * w // 2)
    for i in range(masks.shape[0]):
        assert got[i] == orc.find_contours(masks[i]), i


def test_full_size_and_capacity_overflow():
    masks = np.stack([_blob_mask(7 + s, 512, 512, 4, 0.5) for s in range(3)])
    with binding.Engine(512, 512, max_batch=2) as eng:
        got = eng.extract_contours(masks, cap_points=200000, cap_contours=20000)
        small = eng.extract_contours(masks, cap_points=16, cap_contours=20000)
        few = eng.extract_contours(masks, cap_points=200000, cap_contours=2)
    for i in range(3):
        want = orc.find_contours(masks[i])
        assert got[i] == want and len(want) > 10
        assert small[i] is None and few[i] is None


def test_large_masks_bit_plane_grows_and_falls_back():
    """The tracer walks an LDS bit plane of the mask: 32 KB at 512x512, 128 KB at 1024x1024 (the kernel's dynamic-LDS
    opt-in has to grow inside one process), and masks past the LDS budget (1536x1024 = 192 KB) keep probing global memory."""
    for h, w in ((256, 256), (1024, 1024), (1536, 1024)):
        masks = np.stack([_blob_mask(31 + s, h, w, 6, 0.5) for s in range(2)])
        with binding.Engine(h, w, base=16, levels=2, max_batch=2) as eng:
            got = eng.extract_contours(masks, cap_points=400000, cap_contours=40000)
        for i in range(2):
            assert got[i] == orc.find_contours(masks[i]), (h, w, i)


def test_segment_raw16_whole_device_half_of_the_pipeline():
    """RAW16 -> tile -> UNet -> argmax -> postprocess_mask -> mask_to_image -> contours, all on the device in one call;
    every output equals the oracle chain run stage by stage."""
    from miunet import synth
    from miunet.spec import UNetSpec, pack_weights

    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_threshold_weights(spec))
    # five images through micro-batches of 2, 2, 1: the upload + preprocessing of micro-batch k + 1 runs on the second stream into
    # the other tile buffer while the network works on k (the third micro-batch waits for the first one's readers)
    raws = [synth.make_raw16(1536, 2048, seed=21), synth.make_raw16(600, 800, seed=41), np.full((64, 64), 7, np.uint16),
            synth.make_raw16(700, 900, seed=5), synth.make_raw16(1536, 2048, seed=22)]
    with binding.Engine(512, 512, max_batch=2) as eng:
        eng.load_weights(blob)
        first = eng.segment_raw16(raws)
        eng.segment_raw16(raws[::-1])                       # other data through the same buffers and graphs
        tiles, masks, cont = eng.segment_raw16(raws)        # hipGraph replays by now
        stages = eng.last_stage_ms()
    assert np.array_equal(first[0], tiles) and np.array_equal(first[1], masks) and first[2] == cont
    assert set(stages) == {"upload_preprocess", "network", "postprocess", "contours", "download"}
    assert all(v > 0.0 for v in stages.values()) and stages["network"] > stages["postprocess"]
    n_with_contours = 0
    for i, raw in enumerate(raws):
        tile = orc.preprocess_raw(raw)
        _, labels = orc.unet_forward(blob, tile[None, :, :, None], want_logits=False)
        vis = orc.mask_to_image(orc.postprocess_mask(labels[0]))
        assert np.array_equal(tiles[i], tile) and np.array_equal(masks[i], vis)
        assert cont[i] == orc.find_contours(vis)
        n_with_contours += bool(cont[i])
    assert n_with_contours >= 1


def test_three_tracers_agree_on_adversarial_masks():
    """The device tracer, the host BorderTracer and the oracle's Suzuki-Abe restatement are three independently written
    border followers.  Shapes built to break one -- 1-pixel spurs, diagonal-only links, nested rings touching the frame,
    strokes lying on the frame, a comb, a checkerboard -- and dense unsmoothed noise (all of that at once) must give the same
    point sequences, in the same order, from all three."""
    from miunet import hostlib
    from test_host_cpu import _adversarial_masks

    masks = [m for m in _adversarial_masks()]
    masks[5] = (np.indices((18, 18)).sum(0) % 2 * 255).astype(np.uint8)          # engine sizes are even
    for m in masks:
        h, w = m.shape
        with binding.Engine(h, w, 1, 16, 1, 3, max_batch=1) as eng:
            got = eng.extract_contours(m[None], cap_points=4 * h * w, cap_contours=h * w)[0]
        assert got == orc.find_contours(m) == hostlib.extract_contours(m)
    rng = np.random.default_rng(11)
    # (two width classes of the device tracer: 48 = rows packed back to back in the bit plane; 64 = rows are whole words, the plane
    # carries a zero row above and below and the walk has no row tests -- noise touches every edge of the frame)
    for hh, ww in ((32, 48), (24, 64)):
        for density in (0.25, 0.5, 0.75):
            noise = np.stack([(rng.random((hh, ww)) < density).astype(np.uint8) * 255 for _ in range(8)])
            with binding.Engine(hh, ww, 1, 16, 1, 3, max_batch=4) as eng:
                got = eng.extract_contours(noise, cap_points=4 * hh * ww, cap_contours=hh * ww)
            for i in range(8):
                assert got[i] == orc.find_contours(noise[i]) == hostlib.extract_contours(noise[i]), (hh, ww, density, i)
