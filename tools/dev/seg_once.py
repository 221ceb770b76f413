"""dev: mi_unet_segment_raw16 a few times -- `python tools/dev/seg_once.py [c5]` (c5 = BASELINE config 5: 1024^2 x 3, fp16, 8 images);
run under `rocprofv3 --kernel-trace --stats` for the per-kernel times of the stages around the network"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights
c5 = len(sys.argv) > 1 and sys.argv[1] == "c5"
spec = UNetSpec(in_ch=3, base=32, levels=5) if c5 else UNetSpec()
size, nimg, algo = (1024, 8, "fp16") if c5 else (512, 16, "auto")
blob = pack_weights(spec, synth.make_threshold_weights(spec))
raws = [synth.make_raw16(1536, 2048, seed=100 + i) for i in range(nimg * spec.in_ch)]
with binding.Engine(size, size, in_ch=spec.in_ch, base=spec.base, levels=spec.levels, classes=spec.classes, max_batch=nimg, conv_algo=algo) as eng:
    eng.load_weights(blob)
    for _ in range(6):
        t0 = time.perf_counter(); eng.segment_raw16(raws, 1 << 15, 64); print("segment ms", (time.perf_counter() - t0) * 1e3, eng.last_stage_ms())
