set -o pipefail
mkdir -p gpurun_out/r04e
timeout -k 10 900 python tools/dev/asm_bringup.py > gpurun_out/r04e/asm_bringup.txt 2>&1; rc=$?; grep -v "bad rows\|bad cols\|bad channels" gpurun_out/r04e/asm_bringup.txt
[ $rc -ne 0 ] && exit $rc
python bench.py --per-layer --no-extras --no-cpu-baseline > gpurun_out/r04e/bench_asm.json 2> gpurun_out/r04e/per_layer_asm.txt; echo "bench rc=$?"; tail -c 600 gpurun_out/r04e/bench_asm.json; cat gpurun_out/r04e/per_layer_asm.txt
MIUNET_WINO4_ASM=0 python bench.py --per-layer --no-extras --no-cpu-baseline > gpurun_out/r04e/bench_hipcc.json 2> gpurun_out/r04e/per_layer_hipcc.txt; tail -c 300 gpurun_out/r04e/bench_hipcc.json; tail -3 gpurun_out/r04e/per_layer_hipcc.txt
