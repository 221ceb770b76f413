#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03 [fp32|bf16|fp16|all]
# then, back in the build container, tools/collect_profiles.py r03 copies the judged files into profiles/ with the round tag.
# bench line + per-layer table, rocprofv3 kernel-trace stats of the same bench command, and the PMC passes (FETCH_SIZE,
# WRITE_SIZE, SQ MFMA-busy group: separate runs, kernel-trace only) summarised by profiles/summarize_pmc.py.
# The program sits directly behind `--` (the profiler's preloaded library has initialised the GPU by then: no env/bash hop).
set -e -o pipefail
tag=${1:-r02}
what=${2:-all}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES"

passes() {   # $1 = name, rest = bench arguments
    name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_trace -o run -- python bench.py --verbose --no-cpu-baseline --no-extras "$@" > $out/${name}_bench_under_rocprof.json 2> $out/${name}_rocprof.log
    find $out/${name}_trace -name "*kernel_stats.csv" -exec cp {} $out/${name}_kernel_stats.csv \;
    echo "$name: kernel trace done"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${name}_pmc_fetch -o run -- python bench.py --verbose --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > /dev/null 2>> $out/${name}_rocprof.log
    echo "$name: FETCH_SIZE done"
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${name}_pmc_write -o run -- python bench.py --verbose --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > /dev/null 2>> $out/${name}_rocprof.log
    echo "$name: WRITE_SIZE done"
    rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/${name}_pmc_sq -o run -- python bench.py --verbose --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > /dev/null 2>> $out/${name}_rocprof.log
    echo "$name: SQ done"
    python profiles/summarize_pmc.py --fetch $(find $out/${name}_pmc_fetch -name "*counter_collection.csv") \
        --write $(find $out/${name}_pmc_write -name "*counter_collection.csv") --sq $(find $out/${name}_pmc_sq -name "*counter_collection.csv") \
        --command "python bench.py --verbose --no-cpu-baseline --no-extras --steps 3 --warmup 1 $*" --out $out/pmc_${name}.json > /dev/null
}

if [ "$what" = "fp32" ] || [ "$what" = "all" ]; then
    python bench.py --verbose --per-layer > $out/bench.json 2> $out/per_layer.txt
    echo "bench done"
    passes fp32
    # every profile that carries this round's tag is MEASURED in this round: the one-GPU figure of BASELINE
    # configs[3] (global batch 512 in micro-batches of 16) and the one-rank rehearsal of the RCCL path with the configs[3] strong loop
    python bench.py --global-batch 512 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/global_batch512_bench.json 2> $out/global_batch512.err
    BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 > $out/dist_rehearsal_1rank.json 2> $out/dist_rehearsal.err
    echo "global batch 512 + one-rank RCCL rehearsal done"
    python -m pytest tests/test_gpu_insitu.py tests/test_gpu_numeric_range.py -q -s > $out/insitu_and_numeric_range.txt 2>&1 || echo "in-situ / numeric-range tests FAILED"
    python tools/bench_pipeline.py 2>&1 | grep -v "^Processing\|^Original\|^Scaled\|^Extracted\|^Overlay\|^JSON\|^Total\|^Resources" > $out/pipeline_stages.txt
    echo "parity logs + pipeline stages done"
fi
if [ "$what" = "bf16" ] || [ "$what" = "all" ]; then
    python bench.py --verbose --conv-algo bf16 --batch 128 --steps 5 --no-cpu-baseline --no-extras --per-layer > $out/bf16_bench.json 2> $out/bf16_per_layer.txt
    passes bf16 --conv-algo bf16 --batch 128
fi
if [ "$what" = "fp16" ] || [ "$what" = "all" ]; then     # BASELINE config 5's network
    C5="--conv-algo fp16 --size 1024 --in-ch 3 --base 32 --levels 5 --batch 8 --micro-batch 8"
    python bench.py --verbose $C5 --steps 10 --no-cpu-baseline --no-extras --per-layer > $out/fp16_bench.json 2> $out/fp16_per_layer.txt
    passes fp16 $C5
fi
ls -la $out
