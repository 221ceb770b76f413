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
