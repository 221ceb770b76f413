#!/usr/bin/env python3
"""dev: host timeline of mi_unet_segment_raw16 (MIUNET_RAW_TRACE=1) for 16 RAW images, pageable and pinned"""
import os, sys, time
os.environ["MIUNET_RAW_TRACE"] = "1"
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
    prep = eng.segment_raw16_prepare(raws, 1 << 15, 64)
    for i in range(4):
        print(f"--- pageable call {i}", file=sys.stderr)
        t0 = time.perf_counter(); eng.segment_raw16_run(prep); print(f"total {1e3*(time.perf_counter()-t0):.3f} ms  stages {eng.last_stage_ms()}", file=sys.stderr)
    pins = [binding.PinnedArray(r.shape, np.uint16) for r in raws]
    for pa, r in zip(pins, raws):
        pa.a[...] = r
    prep = eng.segment_raw16_prepare([pa.a for pa in pins], 1 << 15, 64)
    for i in range(3):
        print(f"--- pinned call {i}", file=sys.stderr)
        t0 = time.perf_counter(); eng.segment_raw16_run(prep); print(f"total {1e3*(time.perf_counter()-t0):.3f} ms  stages {eng.last_stage_ms()}", file=sys.stderr)
