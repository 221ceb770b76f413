#!/bin/bash
# Same-card A/B of the 16-bit plans (BASELINE configs 3 and 5) for several environment variants in ONE gpurun call:
# per-layer tables, two rounds, variants interleaved.   usage: tools/dev/ab16.sh <outdir> "<name>:<ENV=V ENV2=V2>" ...
set -e -o pipefail
out=$1; shift
mkdir -p "$out"
C3="--conv-algo bf16 --batch 128 --steps 5 --warmup 2"
C5="--conv-algo fp16 --size 1024 --in-ch 3 --base 32 --levels 5 --batch 8 --micro-batch 8 --steps 10 --warmup 3"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'parity', d['parity']['ok'], 'err', d['parity'].get('max_abs_logit_err'), flush=True)
except Exception as e:
    print(sys.argv[2], 'no bench line:', e, flush=True)
PY
}
for round in 1 2; do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python bench.py $C3 --no-cpu-baseline --no-extras --per-layer > "$out/c3_${name}_r${round}.json" 2> "$out/c3_${name}_r${round}.txt" || echo "  (config3 $name: bench exit code $?)"
    show "$out/c3_${name}_r${round}.json" "config3 $name r$round"
    env $envs python bench.py $C5 --no-cpu-baseline --no-extras --per-layer > "$out/c5_${name}_r${round}.json" 2> "$out/c5_${name}_r${round}.txt" || echo "  (config5 $name: bench exit code $?)"
    show "$out/c5_${name}_r${round}.json" "config5 $name r$round"
  done
done
