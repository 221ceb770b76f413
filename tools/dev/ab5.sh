#!/bin/bash
# Same-card A/B of BASELINE config 5 (fp16, 1024x1024x3, 5 levels, base 32, batch 8): per-layer tables per (name:ENV...) variant.
set -e -o pipefail
out=$1; shift
mkdir -p "$out"
for round in 1 2; do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python bench.py --conv-algo fp16 --size 1024 --in-ch 3 --base 32 --levels 5 --batch 8 --micro-batch 8 --steps 10 --warmup 3 \
        --no-cpu-baseline --no-extras --per-layer > "$out/${name}_r${round}.json" 2> "$out/${name}_r${round}.txt"
    python - "$out/${name}_r${round}.json" "$name r$round" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'parity', d['parity']['ok'])
PY
  done
done
