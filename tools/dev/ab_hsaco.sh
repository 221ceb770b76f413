#!/bin/bash
# Same-card A/B of code objects of the assembly F(4x4,3x3) kernel: the lab library (libmiunet_exp.so) loads MIUNET_WINO4A_HSACO.
# usage: tools/dev/ab_hsaco.sh <outdir> name=file.hsaco ...     (two interleaved rounds of bench.py --per-layer)
set -o pipefail
out=$1; shift
mkdir -p $out
export MIUNET_LIB=$PWD/unet-medical-image-contour-segmentation-cpp_amd/libmiunet_exp.so
for round in 1 2; do
  for v in "$@"; do
    name=${v%%=*}; file=${v#*=}
    MIUNET_WINO4A_HSACO=$PWD/$file python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-layer > $out/${name}_r$round.json 2> $out/${name}_r$round.txt || echo "($name: bench exit code $?)"
    python - $out/${name}_r$round.json $out/${name}_r$round.txt "$name r$round" <<'PY'
import json,sys,re
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=open(sys.argv[2]).read()
w=sum(float(l.split()[2]) for l in t.splitlines() if 'conv3x3_wino4a' in l or ('conv3x3_wino4 ' in l))
print(sys.argv[3], 'images/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'two-block F(4x4) layers', round(w,3), 'ms', 'parity', d['parity']['ok'], flush=True)
PY
  done
done
