#!/bin/bash
# Same-card A/B of the LDS piece layout of the 16-bit kernels (MIUNET_LDS_SWZ=0: round-2 layout, 1: shipped), in ONE gpurun call:
# per-layer tables of BASELINE configs 3 and 5 for both layouts, twice, then the LDS counters of both.
# usage: tools/dev/ab_swz.sh <outdir>
set -e -o pipefail
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C3="--conv-algo bf16 --batch 128 --steps 5 --warmup 2"
C5="--conv-algo fp16 --size 1024 --in-ch 3 --base 32 --levels 5 --batch 8 --micro-batch 8 --steps 10 --warmup 3"
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'parity', d['parity']['ok'], flush=True)
PY
}
for round in 1 2; do
  for swz in 0 1; do
    MIUNET_LDS_SWZ=$swz python bench.py $C3 --no-cpu-baseline --no-extras --per-layer > "$out/c3_swz${swz}_r${round}.json" 2> "$out/c3_swz${swz}_r${round}.txt"
    show "$out/c3_swz${swz}_r${round}.json" "config3 swz=$swz r$round"
    MIUNET_LDS_SWZ=$swz python bench.py $C5 --no-cpu-baseline --no-extras --per-layer > "$out/c5_swz${swz}_r${round}.json" 2> "$out/c5_swz${swz}_r${round}.txt"
    show "$out/c5_swz${swz}_r${round}.json" "config5 swz=$swz r$round"
  done
done
# counters: the program sits directly behind `--`; the switch travels in the environment of rocprofv3 itself
for swz in 0 1; do
  export MIUNET_LDS_SWZ=$swz
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d "$out/pmc_c3_swz$swz" -o run -- python bench.py $C3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$out/pmc_c3_swz$swz.log" || echo "pmc pass swz=$swz failed (see log)"
  python - "$out/pmc_c3_swz$swz" "swz=$swz" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True)
if not f: print(sys.argv[2], 'no counter file'); sys.exit(0)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'].split('<')[0].split('(')[0][-40:]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_LDS_IDX_ACTIVE',0))[:8]:
    a=v.get('SQ_LDS_IDX_ACTIVE',0); c=v.get('SQ_LDS_BANK_CONFLICT',0)
    print(sys.argv[2], f"{k:42s} LDS_IDX_ACTIVE {a:.3e}  BANK_CONFLICT {c:.3e}  ratio {c/a if a else 0:.3f}", flush=True)
PY
done
