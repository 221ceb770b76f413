import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights
spec = UNetSpec()
blob = pack_weights(spec, synth.make_threshold_weights(spec))
raws = [synth.make_raw16(1536, 2048, seed=100 + i) for i in range(16)]
with binding.Engine(512, 512, max_batch=16) as eng:
    eng.load_weights(blob)
    for _ in range(4):
        t0 = time.perf_counter(); eng.segment_raw16(raws, 1 << 15, 64); print("segment ms", (time.perf_counter() - t0) * 1e3)
