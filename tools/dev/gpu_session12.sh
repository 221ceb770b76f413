set -o pipefail
out=gpurun_out/r04p
mkdir -p $out
for round in 1 2; do
for mode in 1 0; do
  MIUNET_WINO4_ASM_B=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-layer > $out/b${mode}_r$round.json 2> $out/b${mode}_r$round.txt
  python - $out/b${mode}_r$round.json "MIUNET_WINO4_ASM_B=$mode r$round" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms', d['parity']['ok'], d['parity']['max_abs_logit_err'])
PY
  grep "up4.c1\|inc.c2\|up4.c2" $out/b${mode}_r$round.txt
done
done
