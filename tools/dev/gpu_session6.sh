set -o pipefail
out=${1:-gpurun_out/r04g}
mkdir -p $out
timeout -k 10 900 python tools/dev/asm_bringup.py > $out/asm_bringup.txt 2>&1; rc=$?; grep -v "bad rows\|bad cols\|bad channels" $out/asm_bringup.txt
[ $rc -ne 0 ] && exit $rc
bash tools/dev/gpu_stamps.sh $out/stamps.txt || exit 1
python bench.py --per-layer --no-extras --no-cpu-baseline > $out/bench_asm.json 2> $out/per_layer_asm.txt; echo "bench rc=$?"; tail -c 250 $out/bench_asm.json; grep "wino4a\|sum of" $out/per_layer_asm.txt
