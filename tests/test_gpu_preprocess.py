"""the RAW16 preprocessing arithmetic on the device, fused in front of the network, bit-exact against
the oracle's restatement of src/preprocess.cpp:65-118."""
import numpy as np
import pytest

import oracle_lib as orc
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_threshold_weights(spec))
    with binding.Engine(512, 512, max_batch=4) as eng:
        eng.load_weights(blob)
        yield eng, blob


def test_device_preprocess_is_bit_exact_and_feeds_the_network(engine):
    eng, blob = engine
    raws = [synth.make_raw16(1536, 2048, seed=77), synth.make_raw16(200, 300, seed=78), synth.make_raw16(512, 512, seed=79),
            synth.make_raw16(700, 333, seed=80), synth.make_raw16(1031, 517, seed=81),      # odd sizes: ragged min/max tail
            np.full((40, 30), 1234, np.uint16),                                             # mn == mx -> all zeros
            np.full((8, 8), 65535, np.uint16),                                              # mn == mx == 65535: u16 wrap
            np.array([[0, 1000], [2000, 3000]], np.uint16),                                 # 2x2 -> 512x512
            np.array([[0, 65535]], np.uint16)]
    tiles, labels, _ = eng.infer_raw16(raws)                  # 9 images through micro-batches of 4, 4, 1
    want_tiles = np.stack([orc.preprocess_raw(r) for r in raws])
    assert np.array_equal(tiles, want_tiles)
    _, want_labels = orc.unet_forward(blob, want_tiles[:3, :, :, None], want_logits=False)
    assert np.array_equal(labels[:3], want_labels)
    # same tiles through the u8 entry point give the same labels (the fused path adds nothing but the preprocessing)
    labels_u8, _ = eng.infer(want_tiles[..., None])
    assert np.array_equal(labels, labels_u8)


def test_raw16_error_paths(engine):
    eng, _ = engine
    with pytest.raises(binding.MiUnetError):
        eng.infer_raw16([np.zeros((0, 5), np.uint16)])


def test_bad_image_late_in_a_call_is_rejected_before_anything_runs_and_the_handle_stays_usable(engine):
    """a bad description in a LATER micro-batch used to be found after the first network was enqueued -- with H2D
    copies still reading the caller's pinned buffers.  Every description is now validated up front (nothing is enqueued), the
    message names the image, and the same handle then gives the same results as before."""
    eng, _ = engine
    raws = [synth.make_raw16(300 + 8 * i, 400, seed=50 + i) for i in range(9)]           # micro-batches of 4, 4, 1
    want = eng.segment_raw16(raws)
    pins = [binding.PinnedArray(r.shape, np.uint16) for r in raws]
    for pa, r in zip(pins, raws):
        pa.a[...] = r
    bad = [pa.a for pa in pins]
    bad[6] = np.zeros((0, 400), np.uint16)                                                # image 6: second micro-batch
    with pytest.raises(binding.MiUnetError) as ei:
        eng.segment_raw16(bad)
    assert ei.value.code == 1 and "image 6" in str(ei.value)
    for pa in pins:                                                                       # nothing in flight reads these any more
        pa.close()
    got = eng.segment_raw16(raws)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
