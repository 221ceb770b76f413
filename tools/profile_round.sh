#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01c
# bench line, per-layer table, rocprofv3 kernel-trace stats of the same bench command, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) summarised by profiles/summarize_pmc.py.
set -e
tag=${1:-r01c}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py --per-layer > $out/bench.json 2> $out/per_layer.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o run -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $out/rocprof.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o run -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $out/rocprof.log
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
python profiles/summarize_pmc.py $(find $out/pmc_fetch -name "*counter_collection.csv") $(find $out/pmc_write -name "*counter_collection.csv") $out/pmc_traffic.json > /dev/null
ls -la $out
