set -o pipefail
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -x -q > gpurun_out/r04a/gpu_suite.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04a/gpu_suite.txt
tail -5 gpurun_out/r04a/gpu_suite.txt
python bench.py --per-layer > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/per_layer.txt; echo "bench rc=$?"
tail -c 1500 gpurun_out/r04a/bench.json
python tools/dev/guard_explore.py > gpurun_out/r04a/guard_explore.txt 2>&1; tail -8 gpurun_out/r04a/guard_explore.txt
