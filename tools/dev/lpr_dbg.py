import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
from miunet import binding
B, H, W, Cin, Cout = 3, 96, 80, 32, 32
r = np.random.default_rng(1)
x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
full = binding.layer_debug("conv3x3_fp16r_lpout", x, w, None, None, relu=True)
ref = orc.maxpool2x2(full)
for it in range(3):
    got = binding.layer_debug("conv3x3_fp16r_pool_lpout", x, w, None, None, relu=True)
    bad = (got != ref)
    print("run", it, "bad values", bad.sum(), "nan", np.isnan(got).sum())
    for b in range(B):
        ys, xs, cs = np.nonzero(bad[b])
        if len(ys):
            print("  image", b, "rows", sorted(set(ys.tolist())), "cols", sorted(set(xs.tolist())), "chans", sorted(set(cs.tolist()))[:40])
            y, xx, c = ys[0], xs[0], cs[0]
            print("   first", (y, xx, c), "got", got[b, y, xx, c], "ref", ref[b, y, xx, c], "window", full[b, 2*y:2*y+2, 2*xx:2*xx+2, c].ravel())
