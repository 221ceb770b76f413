#!/bin/bash
# timing-only builds of the fused first layer (lab library): where does the time of conv3x3_lpr<.., FIRST> go
set -o pipefail
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r04w}; mkdir -p $out
export MIUNET_LIB=$PWD/unet-medical-image-contour-segmentation-cpp_amd/libmiunet_exp.so
export BENCH_TIMING_ONLY=1
C5="--conv-algo fp16 --size 1024 --in-ch 3 --base 32 --levels 5 --batch 8 --micro-batch 8"
for e in 0 4 8 32 40 20; do
  MIUNET_LPR_EXP=$e python bench.py $C5 --steps 10 --no-cpu-baseline --no-extras --per-layer > $out/fp16_exp$e.json 2> $out/fp16_exp$e.txt || true
  echo "fp16 MIUNET_LPR_EXP=$e: $(grep -E '^inc\.c2' $out/fp16_exp$e.txt)"
done
