"""How far does Winograd F(4x4,3x3) hold the path's ABSOLUTE bar (logits within 1e-3 of the fp32 reference) once the weights
leave the He-initialised range?  (VERDICT r02 missing #4 / next #6: the importer exists because real weights are expected,
/root/reference/.gitignore:2-8.)

The 512x512 network is loaded with BatchNorm scales that push the activations to 1e2 - 1e3 and a head scaled to logits of a
given magnitude; the fp32 oracle is the reference.  Reported per weight set: the error of F(4x4,3x3) forced on every layer,
F(2x2,3x3) on every layer, the direct implicit GEMM, and of `auto` with the engine's numeric guard (mi_unet_numeric_guard:
a probe tile through both Winograd plans at load time; F(4x4) is kept only if they agree within 5e-4, half the bar).

What is asserted is the guarantee documented in include/mi_unet.h:
  * `auto` is within 1e-3 of the oracle whenever the guard kept F(4x4);
  * when the guard tripped, `auto` IS the F(2x2) plan (bit-identical logits), i.e. the tightest Winograd form the library has;
  * every algorithm stays within 2e-5 of the logit RANGE (the relative bound that remains meaningful at any magnitude)."""
import numpy as np
import pytest

import oracle_lib as orc
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights

pytestmark = pytest.mark.gpu


def _weights(act_scale, logit_mag, img):
    spec = UNetSpec()
    t = synth.make_weights(spec, 2024)
    t["inc.bn1.gamma"] = (t["inc.bn1.gamma"] * act_scale).astype(np.float32)       # activations ~ act_scale from the first layer on
    t["inc.bn1.beta"] = (t["inc.bn1.beta"] * act_scale).astype(np.float32)
    blob = pack_weights(spec, t)
    ref, _ = orc.unet_forward(blob, img)
    k = logit_mag / float(np.abs(ref - t["outc.b"][None, :, None, None]).max())
    t["outc.w"] = (t["outc.w"] * k).astype(np.float32)
    blob = pack_weights(spec, t)
    ref, lab = orc.unet_forward(blob, img)
    return blob, ref, lab


def _run(blob, img, algo, monkeypatch, guard=None, batch=1):
    if guard is None:
        monkeypatch.delenv("MIUNET_WINO4_GUARD", raising=False)
    else:
        monkeypatch.setenv("MIUNET_WINO4_GUARD", guard)
    monkeypatch.setenv("MIUNET_WINO4_MIN_WG", "0")                                 # F(4x4) on every packed layer at batch 1 too
    with binding.Engine(512, 512, max_batch=batch, conv_algo=algo) as eng:
        eng.load_weights(blob)
        labels, logits = eng.infer(img, want_logits=True)
        return logits, labels, eng.numeric_guard()


@pytest.mark.parametrize("act_scale,logit_mag", [(1.0, 4.0), (300.0, 50.0), (3000.0, 500.0)])
def test_f4x4_range_and_the_numeric_guard(act_scale, logit_mag, monkeypatch):
    img = synth.make_images(1, 512, 512, 1, 0xF44, "blobs")
    blob, ref, ref_lab = _weights(act_scale, logit_mag, img)
    rng = float(np.abs(ref).max())
    err = {}
    lg4, _, _ = _run(blob, img, "winograd", monkeypatch, guard="0")
    lg2, _, _ = _run(blob, img, "winograd", monkeypatch, guard="2")
    lgd, _, _ = _run(blob, img, "direct", monkeypatch)
    lga, laba, (text, tripped, diff) = _run(blob, img, "auto", monkeypatch)
    for name, lg in (("F(4x4) forced", lg4), ("F(2x2) forced", lg2), ("direct", lgd), ("auto + guard", lga)):
        err[name] = float(np.max(np.abs(lg - ref)))
    print(f"\n[activation scale {act_scale:g}, logit range {rng:.3g}] max |logit - oracle|: " +
          ", ".join(f"{k} {v:.3e}" for k, v in err.items()) + f"\n  {text}")
    assert not np.isnan(lga).any()
    for name, e in err.items():
        assert e <= 2e-5 * max(1.0, rng), (name, e, rng)                           # the relative bound, any magnitude
    assert diff >= 0.0 and ("F(4x4,3x3) kept" in text) == (not tripped)
    if tripped:
        assert np.array_equal(lga, lg2)                                            # auto IS the F(2x2) plan for this weight set
    else:
        assert np.array_equal(lga, lg4) and err["auto + guard"] < 1e-3             # F(4x4) kept and inside the absolute bar
    if act_scale == 1.0:
        assert not tripped and err["F(4x4) forced"] < 1e-4                         # the bench's own weight range: never tripped
    if err["F(4x4) forced"] > 1e-3:
        assert tripped, "F(4x4) left the absolute bar on this weight set and the guard did not notice"
    srt = np.sort(ref, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > max(1e-3, 4e-5 * rng)
    assert np.array_equal(laba[safe], ref_lab[safe])


def test_guard_decision_is_shared_by_clones_and_reported(monkeypatch):
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 5))
    img = synth.make_images(2, 64, 64, 1, 9, "blobs")
    monkeypatch.setenv("MIUNET_WINO4_GUARD", "2")
    with binding.Engine(64, 64, max_batch=2) as eng:
        eng.load_weights(blob)
        text, tripped, _ = eng.numeric_guard()
        assert tripped and "F(2x2,3x3)" in text
        with eng.clone() as c2:
            assert c2.numeric_guard()[1]
            assert np.array_equal(c2.infer(img, want_logits=True)[1], eng.infer(img, want_logits=True)[1])
        kernels = set()
        eng.set_profiling(True)
        eng.infer(img)
        kernels = {s["kernel"] for s in eng.kernel_stats()}
        eng.set_profiling(False)
        assert "conv3x3_wino" in kernels and not any(k.startswith("conv3x3_wino4") for k in kernels)
    monkeypatch.delenv("MIUNET_WINO4_GUARD")
    with binding.Engine(64, 64, max_batch=2, conv_algo="bf16") as eng:
        eng.load_weights(blob)
        assert "not applicable" in eng.numeric_guard()[0]


def _lowpass_weights(act_scale, logit_mag, img):
    """a smoothing first block (|w| in inc.*): white noise averages out under it, structure passes -- the weight family where a
    noise-only probe reads LOW"""
    spec = UNetSpec()
    t = synth.make_weights(spec, 2024)
    for k in t:
        if k.startswith("inc.") and k.endswith(".w"):
            t[k] = np.abs(t[k]).astype(np.float32)
    t["inc.bn1.gamma"] = (t["inc.bn1.gamma"] * act_scale).astype(np.float32)
    t["inc.bn1.beta"] = (t["inc.bn1.beta"] * act_scale).astype(np.float32)
    ref, _ = orc.unet_forward(pack_weights(spec, t), img)
    t["outc.w"] = (t["outc.w"] * (logit_mag / float(np.abs(ref - t["outc.b"][None, :, None, None]).max()))).astype(np.float32)
    blob = pack_weights(spec, t)
    ref, _ = orc.unet_forward(blob, img)
    return blob, ref


def test_guard_probes_a_structured_tile_too(monkeypatch):
    """The guard's decision is the LARGER of two probe differences: the seeded-noise tile of round 3 and a structured tile (ramp +
    blobs).  On a smoothing first block the structured tile reads higher than the noise tile (profiles/r04_guard_explore_lowpass.txt:
    1.7e-4 against 1.3e-4 at activations of 300) -- the direction in which a noise-only probe passes weights that real images push
    further; on the He-initialised family the noise tile reads higher.  Whatever the family: the reported difference is the maximum
    of both, it is never below the noise-only guard's (MIUNET_WINO4_GUARD=3), and F(4x4) never leaves the absolute bar on a "blobs"
    image without the guard having tripped."""
    import re
    img = synth.make_images(1, 512, 512, 1, 0xF44, "blobs")
    structured_higher = 0
    for family, cases in ((_lowpass_weights, [(300.0, 50.0), (900.0, 150.0), (1500.0, 250.0)]), (_weights, [(300.0, 50.0), (1300.0, 220.0)])):
        for act, mag in cases:
            out = family(act, mag, img)
            blob, ref = out[0], out[1]
            lg4, _, _ = _run(blob, img, "winograd", monkeypatch, guard="0")
            _, _, (_, trip3, diff3) = _run(blob, img, "auto", monkeypatch, guard="3")
            _, _, (text, trip1, diff1) = _run(blob, img, "auto", monkeypatch)
            m = re.search(r"noise tile ([0-9.e+-]+), structured tile ([0-9.e+-]+)", text)
            assert m, text
            noise, structured = float(m.group(1)), float(m.group(2))
            err4 = float(np.max(np.abs(lg4 - ref)))
            print(f"\n[{family.__name__} act {act:g}] F(4x4) error on blobs {err4:.3e}; noise tile {noise:.3e}, structured tile {structured:.3e}; "
                  f"noise-only guard tripped {trip3}, two-tile guard tripped {trip1}")
            assert abs(diff1 - max(noise, structured)) <= 6e-3 * diff1 and abs(diff3 - noise) <= 6e-3 * diff3      # (the text carries three digits)
            assert diff1 >= diff3 and (trip1 or not trip3)                      # never weaker than the round-3 guard
            assert trip1 == (diff1 > 5e-4)
            if err4 > 1e-3:
                assert trip1, "F(4x4) left the absolute bar on a structured image and the guard did not notice"
            structured_higher += structured > noise
    assert structured_higher >= 1, "no weight set in the scan drove the structured tile harder than the noise tile"
