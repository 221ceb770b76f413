set -o pipefail
out=gpurun_out/r04k
mkdir -p $out
for round in 1 2; do
for mode in 1 2; do
  MIUNET_WINO4_ASM=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-layer > $out/asm${mode}_r$round.json 2> $out/asm${mode}_r$round.txt
  python - $out/asm${mode}_r$round.json "MIUNET_WINO4_ASM=$mode r$round" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms', d['parity']['ok'])
PY
  grep "down1.c1" $out/asm${mode}_r$round.txt
done
done
