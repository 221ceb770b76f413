#!/usr/bin/env python3
"""Copy the judged artefacts of tools/profile_round.sh from gpurun_out/<tag>/ (scratch) into profiles/ (tracked) as
<tag>_<name>.  Only files the script wrote in THIS round exist under gpurun_out/<tag>/, so nothing can inherit a tag it was not
measured under."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"bench.json": "bench.json", "per_layer.txt": "per_layer.txt", "fp32_kernel_stats.csv": "kernel_stats.csv",
         "fp32_bench_under_rocprof.json": "bench_under_rocprof.json", "pmc_fp32.json": "pmc_fp32.json",
         "global_batch512_bench.json": "global_batch512_bench.json", "dist_rehearsal_1rank.json": "dist_rehearsal_1rank.json",
         "insitu_and_numeric_range.txt": "insitu_and_numeric_range.txt", "pipeline_stages.txt": "pipeline_stages.txt",
         "bf16_bench.json": "bf16_batch128_bench.json", "bf16_per_layer.txt": "bf16_per_layer.txt",
         "bf16_kernel_stats.csv": "bf16_kernel_stats.csv", "pmc_bf16.json": "pmc_bf16.json",
         "fp16_bench.json": "fp16_config5_bench.json", "fp16_per_layer.txt": "fp16_per_layer.txt",
         "fp16_kernel_stats.csv": "fp16_kernel_stats.csv", "pmc_fp16.json": "pmc_fp16.json"}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    n = 0
    for name, dst in NAMES.items():
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copyfile(p, os.path.join(ROOT, "profiles", f"{tag}_{dst}"))
            n += 1
    print(f"copied {n} files from {src} into profiles/ as {tag}_*")


if __name__ == "__main__":
    main()
