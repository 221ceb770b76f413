#!/bin/bash
cd unet-medical-image-contour-segmentation-cpp_amd
for v in NONE NO_STORE NO_EPI; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 -DW4_ABL_$v -c csrc/conv_wino4.hip -o build/conv_wino4.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmiunet.so build/conv_direct.o build/conv_lp.o build/conv_wino.o build/conv_wino4.o build/layers_mem.o build/image_stages.o build/engine.o
  echo "=== $v"
  (cd .. && timeout -k 10 120 python bench.py --per-layer --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -E "inc.c2|down1|up4.c|up3.c2|up1.c1|sum of")
done
