#!/bin/bash
# Same-card A/B: bench.py per-layer tables for several (library, env) variants in ONE gpurun call.
# usage: tools/dev/ab.sh <outdir> <algo fp32|bf16> "<name>:<ENV=V ENV2=V2>" ...
set -e -o pipefail
out=$1; algo=$2; shift 2
mkdir -p "$out"
extra=""; [ "$algo" = bf16 ] && extra="--conv-algo bf16 --batch 128"
for round in 1 2; do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python bench.py $extra --steps 10 --warmup 3 --no-cpu-baseline --no-extras --per-layer \
        > "$out/${name}_r${round}.json" 2> "$out/${name}_r${round}.txt" || echo "  ($name: bench exit code $? -- expected for timing-only experiment builds whose results are wrong)"
    python - "$out/${name}_r${round}.json" "$name r$round" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'parity', d['parity']['ok'])
PY
  done
done
