"""End-to-end parity of the HIP path (through the C-ABI) against the committed golden vectors and the oracle."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights
from test_oracle_unet import load_case

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3          # BASELINE.json north_star: "logits within 1e-3 fp32"


def check_parity(labels, logits, ref_logits, ref_labels=None):
    """north_star bar: logits within 1e-3; label maps identical -- a mismatch is tolerated ONLY where the oracle's
    top-2 margin is itself below the logit tolerance (fp32 re-association), and is counted."""
    assert np.max(np.abs(logits - ref_logits)) < LOGIT_TOL
    # the device argmax follows the device's own logits exactly (first max wins)
    assert np.array_equal(labels, orc_argmax_batch(logits))
    if ref_labels is None:
        ref_labels = orc_argmax_batch(ref_logits)
    srt = np.sort(ref_logits, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    bad = labels != ref_labels
    assert not (bad & (margin > LOGIT_TOL)).any()
    return int(bad.sum())


def same_image_other_batch(labels_a, logits_a, labels_b, logits_b):
    """An image's result must not depend on its batch neighbours.  With MIUNET_SPLITK=0 that holds bit for bit; by
    default the Winograd launcher may cut K into slices when a micro-batch alone cannot fill the chip (split-K is chosen
    from the grid size, i.e. from the micro-batch size), which re-associates the fp32 sums: then the two results agree to
    fp32 rounding and the labels agree wherever the top-2 margin is not itself at rounding level."""
    if np.array_equal(logits_a, logits_b):
        assert np.array_equal(labels_a, labels_b)
        return
    assert np.max(np.abs(logits_a - logits_b)) < 5e-5
    srt = np.sort(logits_a, axis=0)
    safe = (srt[-1] - srt[-2]) > 1e-3
    assert np.array_equal(labels_a[safe], labels_b[safe])


def orc_argmax_batch(logits):
    return np.stack([orc.argmax_planar(l) for l in logits])


def _algo(algo, monkeypatch):
    """"wino4" = the default algorithm with the F(4x4,3x3) kernel forced onto every eligible layer whatever its grid size
    (by default it only takes layers whose grid fills the chip, i.e. none at these test sizes)."""
    if algo == "wino4":
        monkeypatch.setenv("MIUNET_WINO4_MIN_WG", "1")
        return "winograd"
    return algo


@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd16", "wino4"])
@pytest.mark.parametrize("name", ["unet_b64_l4_64", "unet_b64_l4_48x80", "unet_b16_l3_40x24", "unet_b32_l5_c3_64"])
def test_against_golden(golden_dir, name, algo, monkeypatch):
    algo = _algo(algo, monkeypatch)
    spec, blob, imgs, want = load_case(os.path.join(golden_dir, name + ".npz"))
    b, h, w, _ = imgs.shape
    with binding.Engine(h, w, spec.in_ch, spec.base, spec.levels, spec.classes, max_batch=2, conv_algo=algo) as eng:
        eng.load_weights(blob)
        labels, logits = eng.infer(imgs, want_logits=True)
    flips = check_parity(labels, logits, want)
    assert flips == 0


@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd16", "wino4"])
def test_against_oracle_128_batch_and_microbatching(algo, monkeypatch):
    algo = _algo(algo, monkeypatch)
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 4321))
    imgs = synth.make_images(5, 128, 128, 1, 0xBEEF, "blobs")
    ref_logits, ref_labels = orc.unet_forward(blob, imgs)
    with binding.Engine(128, 128, max_batch=2, conv_algo=algo) as eng:      # 5 images through micro-batches of 2,2,1
        eng.load_weights(blob)
        labels, logits = eng.infer(imgs, want_logits=True)
        flips = check_parity(labels, logits, ref_logits, ref_labels)
        assert flips <= 2
        # labels-only call gives the same labels; single-image calls give the same bits as the batched call
        labels2, none = eng.infer(imgs, want_logits=False)
        assert none is None and np.array_equal(labels, labels2)
        l1, g1 = eng.infer(imgs[3:4], want_logits=True)
        same_image_other_batch(l1[0], g1[0], labels[3], logits[3])


@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd16"])
def test_full_size_512_one_image_vs_oracle_and_batch16_properties(algo):
    """configs[1] of BASELINE.json: batch 16 at 512x512.  The oracle checks one image in full; the other 15 are
    covered by size-independent properties: batch independence (bit-identical to the single-image run), and
    determinism across two runs."""
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 1234))
    imgs = synth.make_images(16, 512, 512, 1, 0x5EED, "bytes")
    with binding.Engine(512, 512, max_batch=16, conv_algo=algo) as eng:
        eng.load_weights(blob)
        labels, logits = eng.infer(imgs, want_logits=True)
        labels_b, logits_b = eng.infer(imgs, want_logits=True)
        assert np.array_equal(labels, labels_b) and np.array_equal(logits, logits_b)
        l7, g7 = eng.infer(imgs[7:8], want_logits=True)
        same_image_other_batch(l7[0], g7[0], labels[7], logits[7])
    ref_logits, ref_labels = orc.unet_forward(blob, imgs[7:8])
    flips = check_parity(labels[7:8], logits[7:8], ref_logits, ref_labels)
    assert flips <= 8
    assert set(np.unique(labels)) <= {0, 1, 2}


def test_error_paths():
    with binding.Engine(64, 64, max_batch=1) as eng:
        with pytest.raises(binding.MiUnetError) as ei:
            eng.infer(np.zeros((1, 64, 64, 1), np.uint8))
        assert ei.value.code == 5 and "Engine not initialized" in str(ei.value)     # src/process.cpp:195
        with pytest.raises(binding.MiUnetError) as ei:
            eng.load_weights(b"not a weight file")
        assert ei.value.code == 4
        spec = UNetSpec(1, 32, 4, 3)
        with pytest.raises(binding.MiUnetError):
            eng.load_weights(pack_weights(spec, synth.make_weights(spec, 1)))       # topology mismatch
        with pytest.raises(binding.MiUnetError) as ei:
            eng.load_weights("/nonexistent/engine.miw")
        assert "Engine file not found" in str(ei.value)                              # src/initialize.cpp:43
        with pytest.raises(ValueError):
            eng.infer(np.zeros((1, 32, 32, 1), np.uint8))                             # src/process.cpp:126-128
    with pytest.raises(binding.MiUnetError):
        binding.Engine(100, 64)                                                      # not a multiple of 2^levels


@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd16"])
def test_fused_pooling_is_bit_identical_to_the_pool_kernel(algo, monkeypatch):
    """The conv epilogues write the 2x2-max-pooled tensor themselves; MIUNET_FUSE_POOL=0 runs the stand-alone pooling
    kernel instead.  max and (+shift, ReLU) commute exactly, so logits must agree bit for bit."""
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 77))
    imgs = synth.make_images(2, 96, 160, 1, 0x1111, "blobs")
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIUNET_FUSE_POOL", flag)
        with binding.Engine(96, 160, max_batch=2, conv_algo=algo) as eng:
            eng.load_weights(blob)
            eng.set_profiling(True)
            labels, logits = eng.infer(imgs, want_logits=True)
            kernels = [s["kernel"] for s in eng.kernel_stats()]
        assert ("maxpool2x2" in kernels) == (flag == "0")
        outs.append((labels, logits))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_wino4_fusions_on_small_grids(monkeypatch):
    """The F(4x4,3x3) kernel normally takes a layer only when its grid fills the chip; MIUNET_WINO4_MIN_WG=1 forces it onto
    small test images so its own fusions are checked against the stand-alone kernels: pooled maxima from the epilogue
    (bit-identical: max and +shift/ReLU commute), the 1x1 head + argmax in the last conv's epilogue (same products, a
    different summation order: logits to 1e-5, labels wherever the top-2 margin exceeds that), the persistent multi-tile
    walk of the one-block variant (5 x 64 tiles > 256 CUs), and all of it against the oracle."""
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 78))
    imgs = synth.make_images(5, 128, 128, 1, 0x2222, "blobs")
    ref_logits, ref_labels = orc.unet_forward(blob, imgs)
    monkeypatch.setenv("MIUNET_WINO4_MIN_WG", "1")
    outs = {}
    for pool, head in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("MIUNET_FUSE_POOL", pool)
        monkeypatch.setenv("MIUNET_FUSE_HEAD", head)
        with binding.Engine(128, 128, max_batch=5) as eng:
            eng.load_weights(blob)
            eng.set_profiling(True)
            labels, logits = eng.infer(imgs, want_logits=True)
            kernels = [s["kernel"] for s in eng.kernel_stats()]
        assert ("maxpool2x2" in kernels) == (pool == "0")
        assert ("head_argmax" in kernels) == (head == "0") and ("conv3x3_wino4+head" in kernels) == (head == "1")
        assert "conv3x3_wino" not in kernels and "conv3x3_wino4" in kernels
        check_parity(labels, logits, ref_logits, ref_labels)
        outs[(pool, head)] = (labels, logits)
    assert np.array_equal(outs[("1", "1")][1], outs[("0", "1")][1]) and np.array_equal(outs[("1", "1")][0], outs[("0", "1")][0])
    la, ga = outs[("1", "1")]
    lb, gb = outs[("1", "0")]
    assert np.max(np.abs(ga - gb)) < 1e-5
    srt = np.sort(gb, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert np.array_equal(la[clear], lb[clear])


def test_graph_replay_matches_eager(monkeypatch):
    """The forward pass is captured into a hipGraph on the second call of a (buffers, batch) key and replayed afterwards
    (the reference replays a CUDA graph, src/process.cpp:147); results must be bit-identical to eager launches."""
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 5))
    imgs = synth.make_images(3, 64, 96, 1, 0x2222, "blobs")
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("MIUNET_GRAPH", flag)
        with binding.Engine(64, 96, max_batch=2) as eng:
            eng.load_weights(blob)
            runs = [eng.infer(imgs, want_logits=True) for _ in range(4)]     # eager, capture+launch, replay, replay
            for lab, lg in runs[1:]:
                assert np.array_equal(lab, runs[0][0]) and np.array_equal(lg, runs[0][1])
            eng.load_weights(blob)                                             # reload invalidates the captured graphs
            lab, lg = eng.infer(imgs, want_logits=True)
            assert np.array_equal(lg, runs[0][1])
            res[flag] = runs[0]
    assert np.array_equal(res["0"][1], res["1"][1]) and np.array_equal(res["0"][0], res["1"][0])


def test_config5_shape_1024_three_channels_five_levels():
    """BASELINE config 5's topology (1024x1024x3 input, 5 levels, base 32) through the fp32 engine: one image against the
    oracle in full, a second one in the batch for independence.  (Its fp16 arithmetic is not built yet: DESIGN.md §9.)"""
    spec = UNetSpec(3, 32, 5, 3)
    blob = pack_weights(spec, synth.make_weights(spec, 555))
    imgs = synth.make_images(2, 1024, 1024, 3, 0xC5, "blobs")
    with binding.Engine(1024, 1024, 3, 32, 5, 3, max_batch=2) as eng:
        eng.load_weights(blob)
        labels, logits = eng.infer(imgs, want_logits=True)
        l1, g1 = eng.infer(imgs[1:2], want_logits=True)
    same_image_other_batch(l1[0], g1[0], labels[1], logits[1])
    ref_logits, ref_labels = orc.unet_forward(blob, imgs[0:1])
    flips = check_parity(labels[0:1], logits[0:1], ref_logits, ref_labels)
    assert flips <= 16


def test_split_k_is_deterministic_and_optional(monkeypatch):
    """Single images at 512x512 leave the deep levels with 64-128 workgroups for 256 CUs, so the Winograd launcher cuts K
    into slices (slabs summed in slice order by a second kernel: run-to-run deterministic).  MIUNET_SPLITK=0 switches it
    off; both settings meet the parity bar, and without split-K batch independence is bit-exact."""
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 1234))
    imgs = synth.make_images(3, 512, 512, 1, 0x5EED, "bytes")
    ref_logits, ref_labels = orc.unet_forward(blob, imgs[1:2])
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MIUNET_SPLITK", flag)
        with binding.Engine(512, 512, max_batch=4) as eng:
            eng.load_weights(blob)
            eng.set_profiling(True)
            l1, g1 = eng.infer(imgs[1:2], want_logits=True)
            n_launch = len(eng.kernel_stats())
            eng.set_profiling(False)
            l1b, g1b = eng.infer(imgs[1:2], want_logits=True)
            l3, g3 = eng.infer(imgs, want_logits=True)
        assert np.array_equal(g1, g1b) and np.array_equal(l1, l1b)              # deterministic
        check_parity(l1, g1, ref_logits, ref_labels)
        same_image_other_batch(l1[0], g1[0], l3[1], g3[1])
        if flag == "0":
            assert np.array_equal(g1[0], g3[1])                                 # bit-exact batch independence
        out[flag] = g1
    assert np.max(np.abs(out["0"] - out["1"])) < 5e-5


def test_staged_one_block_kernel_in_the_whole_network(monkeypatch):
    """MIUNET_WINO4S=2 sends every one-block F(4x4) launch (the Cout = 64 layers incl. the fused pooling of inc.c2 and the
    fused head of the last conv) to the two-workgroups-per-CU kernel of conv_wino4s.hip, =0 to the persistent kernel: same
    arithmetic in a different schedule, so the logits agree to fma-contraction noise and both meet the 1e-3 bar (ragged
    80x48 blocks, batch 3)."""
    spec = UNetSpec(1, 64, 2, 3)
    blob = pack_weights(spec, synth.make_weights(spec, 99))
    imgs = synth.make_images(3, 80, 48, 1, 0x77, "blobs")
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("MIUNET_WINO4S", mode)
        monkeypatch.setenv("MIUNET_SPLITK", "0")          # same launch family for every layer in both runs
        with binding.Engine(80, 48, 1, 64, 2, 3, max_batch=3) as eng:
            eng.load_weights(blob)
            out[mode] = eng.infer(imgs, want_logits=True)
    assert np.max(np.abs(out["0"][1] - out["2"][1])) < 1e-4
    ref_logits, ref_labels = orc.unet_forward(blob, imgs)
    srt = np.sort(ref_logits, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3
    for mode in ("0", "2"):
        assert np.max(np.abs(out[mode][1] - ref_logits)) < 1e-3
        assert np.array_equal(out[mode][0][safe], ref_labels[safe])
        assert np.array_equal(out[mode][0], np.stack([orc.argmax_planar(l) for l in out[mode][1]]))
