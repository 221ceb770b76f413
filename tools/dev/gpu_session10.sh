set -o pipefail
out=gpurun_out/r04n
mkdir -p $out
python -m pytest tests/test_gpu_insitu.py tests/test_gpu_unet.py -x -q -k "fp32 or unet" > $out/tests.txt 2>&1; rc=$?; tail -6 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
for round in 1 2; do
for mode in 1 0; do
  MIUNET_FUSE_FIRST=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-layer > $out/ff${mode}_r$round.json 2> $out/ff${mode}_r$round.txt
  python - $out/ff${mode}_r$round.json "MIUNET_FUSE_FIRST=$mode r$round" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms', d['parity'])
PY
  grep "inc.c" $out/ff${mode}_r$round.txt
done
done
