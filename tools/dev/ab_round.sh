#!/bin/bash
# Same-card A/B of two builds of libmiunet.so (e.g. the tree a round started from against the final one): the three bench workloads
# and the one-call RAW pipeline, two interleaved rounds.   usage: tools/dev/ab_round.sh <outdir> <name>:<lib.so> <name>:<lib.so> ...
set -o pipefail
out=$1; shift
mkdir -p "$out"
C5="--conv-algo fp16 --size 1024 --in-ch 3 --base 32 --levels 5 --batch 8 --micro-batch 8"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms', 'parity', d['parity']['ok'], flush=True)
except Exception as e:
    print(sys.argv[2], 'no bench line:', e, flush=True)
PY
}
for round in 1 2; do
  for v in "$@"; do
    name=${v%%:*}; lib=${v#*:}
    MIUNET_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/fp32_${name}_r$round.json" 2> "$out/fp32_${name}_r$round.err"; show "$out/fp32_${name}_r$round.json" "configs[1] fp32 $name r$round"
    MIUNET_LIB=$lib python bench.py --conv-algo bf16 --batch 128 --steps 5 --no-cpu-baseline --no-extras > "$out/bf16_${name}_r$round.json" 2> "$out/bf16_${name}_r$round.err"; show "$out/bf16_${name}_r$round.json" "configs[2] bf16 $name r$round"
    MIUNET_LIB=$lib python bench.py $C5 --steps 10 --no-cpu-baseline --no-extras > "$out/fp16_${name}_r$round.json" 2> "$out/fp16_${name}_r$round.err"; show "$out/fp16_${name}_r$round.json" "configs[4] fp16 $name r$round"
    MIUNET_LIB=$lib python tools/dev/seg_once.py 2>&1 | grep "segment ms" | tail -1 | cut -c1-200 | sed "s/^/one-call 512^2 x16 $name r$round: /"
    MIUNET_LIB=$lib python tools/dev/seg_once.py c5 2>&1 | grep "segment ms" | tail -1 | cut -c1-200 | sed "s/^/one-call config 5 $name r$round: /"
  done
done
