#!/bin/bash
# Same-card A/B of code objects of conv3x3_wino4b_f32 (lab library, MIUNET_WINO4B_HSACO); usage: ab_hsaco_b.sh <outdir> name=file.hsaco ...
set -o pipefail
out=$1; shift
mkdir -p $out
export MIUNET_LIB=$PWD/unet-medical-image-contour-segmentation-cpp_amd/libmiunet_exp.so
for round in 1 2; do
  for v in "$@"; do
    name=${v%%=*}; file=${v#*=}
    MIUNET_WINO4B_HSACO=$PWD/$file python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-layer > $out/${name}_r$round.json 2> $out/${name}_r$round.txt || echo "($name: bench exit code $?)"
    python - $out/${name}_r$round.json $out/${name}_r$round.txt "$name r$round" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
w=[l.split()[2] for l in open(sys.argv[2]) if 'conv3x3_wino4b' in l]
print(sys.argv[3], 'images/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'wino4b layers', w, 'parity', d['parity']['ok'], flush=True)
PY
  done
done
