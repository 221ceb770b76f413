set -o pipefail
out=gpurun_out/r04s
mkdir -p $out
ASM_OP=conv3x3_wino4b timeout -k 10 600 python tools/dev/asm_bringup.py 1,16,32,64,64,exact 2,32,64,128,64 1,32,32,64,64,pool 16,128,128,64,64 4,64,64,128,128,pool > $out/asm_bringup.txt 2>&1; rc=$?; grep -v "bad rows\|bad cols\|bad channels" $out/asm_bringup.txt | tail -2
[ $rc -ne 0 ] && exit $rc
for round in 1 2; do
for mode in "1" "0"; do
  MIUNET_FUSE_FIRST=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-layer > $out/ff${mode}_r$round.json 2> $out/ff${mode}_r$round.txt
  python - $out/ff${mode}_r$round.json "MIUNET_FUSE_FIRST=$mode r$round" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms', d['parity']['ok'], d['parity']['max_abs_logit_err'])
PY
  grep "inc.c\|up4.c1" $out/ff${mode}_r$round.txt
done
done
