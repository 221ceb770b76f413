#!/usr/bin/env python3
"""dev: wall time of mi_unet_segment_raw16 for 16 RAW images under MIUNET_RAW_SPLIT (set by the caller), pageable and pinned"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))
import numpy as np
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights
spec = UNetSpec()
blob = pack_weights(spec, synth.make_threshold_weights(spec))
raws = [synth.make_raw16(1536, 2048, seed=100 + i) for i in range(16)]
with binding.Engine(512, 512, max_batch=16) as eng:
    eng.load_weights(blob)
    pins = [binding.PinnedArray(r.shape, np.uint16) for r in raws]
    for pa, r in zip(pins, raws):
        pa.a[...] = r
    for name, imgs in (("pageable", raws), ("pinned", [pa.a for pa in pins])):
        prep = eng.segment_raw16_prepare(imgs, 1 << 15, 64)
        for _ in range(3):
            eng.segment_raw16_run(prep)
        t0 = time.perf_counter()
        for _ in range(8):
            eng.segment_raw16_run(prep)
        dt = (time.perf_counter() - t0) / 8
        st = eng.last_stage_ms()
        print(f"MIUNET_RAW_SPLIT={os.environ.get('MIUNET_RAW_SPLIT', 'default')} {name}: {dt * 1e3:.2f} ms = {16 / dt:.0f} images/s; stages " +
              ", ".join(f"{k} {v:.2f}" for k, v in st.items()))
