set -o pipefail
out=gpurun_out/r04h
mkdir -p $out
bash tools/dev/ab_hsaco.sh $out/ab v1_double=tools/dev/asm_probes/wino4a_v1_double.hsaco v2_triple=tools/dev/asm_probes/wino4a_v2_triple.hsaco || exit 1
bash tools/dev/gpu_stamps_multi.sh $out "base:" "nouload:--timing-only nouload" "novread:--timing-only novread" "notransform:--timing-only notransform" "bare:--timing-only nouload,novread,notransform,nodma" > $out/multi.log 2>&1
grep -A3 "round 2" $out/stamps.txt | grep -v "^--\|wave 2" | cut -c1-250
