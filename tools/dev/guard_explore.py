#!/usr/bin/env python3
"""Where does a white-noise probe pass a weight set that a structured image pushes past 1e-3 under F(4x4,3x3)?  Prints, per
(activation scale, logit magnitude): the F(4x4) / F(2x2) errors of a "blobs" image against the fp32 oracle and the decisions of the
noise-only guard (MIUNET_WINO4_GUARD=3, round 3's) and of the two-tile guard.  Input to tests/test_gpu_numeric_range.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"), os.path.join(ROOT, "tests")]
import oracle_lib as orc  # noqa: E402
from miunet import binding, synth  # noqa: E402
from miunet.spec import UNetSpec, pack_weights  # noqa: E402

os.environ["MIUNET_WINO4_MIN_WG"] = "0"
img = synth.make_images(1, 512, 512, 1, 0xF44, "blobs")
spec = UNetSpec()
for act, mag in [(float(a), float(m)) for a, m in (x.split(":") for x in (sys.argv[1:] or ["300:50", "600:100", "900:150", "1300:220", "1800:300", "2400:400"]))]:
    t = synth.make_weights(spec, 2024)
    if os.environ.get("GUARD_LOWPASS") == "1":           # a smoothing first layer: white noise averages out, structure passes
        for k in t:
            if k.startswith("inc.") and k.endswith(".w"):
                t[k] = np.abs(t[k]).astype(np.float32)
    t["inc.bn1.gamma"] = (t["inc.bn1.gamma"] * act).astype(np.float32)
    t["inc.bn1.beta"] = (t["inc.bn1.beta"] * act).astype(np.float32)
    ref, _ = orc.unet_forward(pack_weights(spec, t), img)
    t["outc.w"] = (t["outc.w"] * (mag / float(np.abs(ref - t["outc.b"][None, :, None, None]).max()))).astype(np.float32)
    blob = pack_weights(spec, t)
    ref, _ = orc.unet_forward(blob, img)
    res = {}
    for name, g in (("f4", "0"), ("f2", "2"), ("noise_guard", "3"), ("two_tile_guard", "1")):
        os.environ["MIUNET_WINO4_GUARD"] = g
        with binding.Engine(512, 512, max_batch=1, conv_algo="auto") as eng:
            eng.load_weights(blob)
            _, lg = eng.infer(img, want_logits=True)
            res[name] = (float(np.max(np.abs(lg - ref))), eng.numeric_guard())
    print(f"act {act:g} logits {float(np.abs(ref).max()):.4g}: F4 err {res['f4'][0]:.3e} F2 err {res['f2'][0]:.3e} | noise-only guard: tripped={res['noise_guard'][1][1]} "
          f"diff={res['noise_guard'][1][2]:.3e} | two-tile: tripped={res['two_tile_guard'][1][1]} :: {res['two_tile_guard'][1][0]}", flush=True)
