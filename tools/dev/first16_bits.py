#!/usr/bin/env python3
"""The 16-bit plans with the first layer fused into inc.c2's loader against the same plans with it as a launch of its own: the
logits must be the same BITS (the fused loader runs conv3x3_first_mfma's arithmetic).  Run on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))
from miunet import binding, synth                      # noqa: E402
from miunet.spec import UNetSpec, pack_weights         # noqa: E402

bad = 0
C5 = UNetSpec(in_ch=3, base=32, levels=5)
for algo, spec, size, batch in (("fp16", C5, 1024, 8), ("fp16", C5, 1024 - 64, 8), ("bf16", C5, 1024, 4), ("bf16", UNetSpec(in_ch=3, base=32, levels=3), 544, 16)):
    blob = pack_weights(spec, synth.make_weights(spec, 77))
    imgs = synth.make_images(batch, size, size, spec.in_ch, 5, "blobs")
    res = {}
    for mode in ("0", "1"):
        os.environ["MIUNET_FUSE_FIRST"] = mode
        with binding.Engine(size, size, in_ch=spec.in_ch, base=spec.base, levels=spec.levels, classes=spec.classes, max_batch=batch, conv_algo=algo) as eng:
            eng.load_weights(blob)
            eng.set_profiling(True)
            labels, logits = eng.infer(imgs, want_logits=True)
            names = [k["kernel"] for k in eng.kernel_stats()]
            res[mode] = (labels.copy(), logits.copy(), names)
    same = np.array_equal(res["0"][1].view(np.uint32), res["1"][1].view(np.uint32))
    d = float(np.abs(res["0"][1] - res["1"][1]).max())
    print(f"{algo} {size}x{size}x{spec.in_ch} batch {batch}: logits bit-identical {same} (max |diff| {d:.3e}), labels equal {np.array_equal(res['0'][0], res['1'][0])}; "
          f"NaN {int(np.isnan(res['1'][1]).sum())}; fused launch present: {any('+first' in n for n in res['1'][2])} / {any('+first' in n for n in res['0'][2])}")
    bad += 0 if same else 1
sys.exit(1 if bad else 0)
