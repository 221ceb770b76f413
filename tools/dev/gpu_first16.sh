#!/bin/bash
# fused first layer of the 16-bit plans: bit equality against the unfused route, the in-situ tests, same-card A/B per layer
set -o pipefail
cd "$(dirname "$0")/../.."
out=gpurun_out/${1:-r04u}; mkdir -p $out
timeout -k 10 300 python tools/dev/first16_bits.py > $out/bits.txt 2>&1 || { tail -20 $out/bits.txt; exit 1; }
cat $out/bits.txt
timeout -k 10 900 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_bf16.py -x -q -m gpu -k "bf16 or fp16" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -3 $out/tests.txt
C5="--conv-algo fp16 --size 1024 --in-ch 3 --base 32 --levels 5 --batch 8 --micro-batch 8"
for round in 1 2; do
  for mode in 1 0; do
    MIUNET_FUSE_FIRST=$mode python bench.py --conv-algo bf16 --batch 128 --steps 5 --no-cpu-baseline --no-extras --per-layer > $out/bf16_ff${mode}_r$round.json 2> $out/bf16_ff${mode}_r$round.txt || exit 1
    MIUNET_FUSE_FIRST=$mode python bench.py $C5 --steps 10 --no-cpu-baseline --no-extras --per-layer > $out/fp16_ff${mode}_r$round.json 2> $out/fp16_ff${mode}_r$round.txt || exit 1
    for k in bf16 fp16; do
      python - $out/${k}_ff${mode}_r$round.json "$k MIUNET_FUSE_FIRST=$mode r$round" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"], 1), "images/s", round(d["ms_per_step"], 3), "ms", d["parity"]["ok"])
PY
      grep -E "^inc\." $out/${k}_ff${mode}_r$round.txt
    done
  done
done
