#!/usr/bin/env python3
"""dev: why does process_single_image's network stage differ between runs?  Prints the facade's guard line and stage lines."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))
from miunet import binding, hostlib, synth
from miunet.spec import UNetSpec, pack_weights
spec = UNetSpec()
blob = pack_weights(spec, synth.make_threshold_weights(spec))
raws = [synth.make_raw16(1536, 2048, seed=100 + i) for i in range(8)]
if len(sys.argv) > 1 and sys.argv[1] == "batch_first":
    pass
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "engine")); wp = os.path.join(d, "engine", "u.miw"); open(wp, "wb").write(blob)
    paths = []
    for i, r in enumerate(raws):
        p = os.path.join(d, f"i{i}.raw"); r.tofile(p); paths.append(p)
    od = os.path.join(d, "out"); os.makedirs(od)
    os.environ["MEDSEG_DEVICES"] = "1"
    assert hostlib.initialize_engine(wp, os.path.join(d, "log"))
    if len(sys.argv) > 1 and sys.argv[1] == "batch_first":
        hostlib.process_image_batch(paths, [2048] * 8, [1536] * 8, od)
        hostlib.process_image_batch(paths, [2048] * 8, [1536] * 8, od)
    for p in paths:
        hostlib.process_single_image(p, 2048, 1536, od)
    t0 = time.perf_counter()
    for p in paths:
        hostlib.process_single_image(p, 2048, 1536, od)
    print("single image ms:", (time.perf_counter() - t0) / 8 * 1e3, file=sys.stderr)
    log = open(hostlib.get_log_path()).read()
    for l in log.splitlines():
        if "numeric guard" in l or "Stage times" in l or "context created" in l:
            print(l, file=sys.stderr)
    hostlib.cleanup_resources()
