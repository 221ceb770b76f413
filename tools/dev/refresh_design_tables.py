#!/usr/bin/env python3
"""Refresh the measured tables of DESIGN.md in place from profiles/<tag>_*: every region between
`<!-- table:NAME -->` and `<!-- /table -->` is regenerated; prose is never touched.

    python tools/dev/refresh_design_tables.py r04
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LINE = re.compile(r"(\S+)\s+(\S+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$")


def layer_rows(path):
    for ln in open(path):
        m = LINE.match(ln)
        if m:
            yield m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))


def fp32_layers(tag):
    out = ["| layer | kernel | ms / launch | algorithmic TFLOP/s | executed ÷ 157.3 TF/s | algorithmic GB/s |", "|---|---|---|---|---|---|"]
    for name, k, ms, tf, gb in layer_rows(f"{ROOT}/profiles/{tag}_per_layer.txt"):
        red = 4.0 if "wino4" in k else 2.25 if "wino" in k else 1.0
        out.append(f"| {name} | {k} | {ms:.3f} | {tf:.1f} | {tf / red / 157.3:.2f} | {gb} |")
    return "\n".join(out)


def lp_layers(tag, which):
    out = ["| layer | kernel | ms / launch | TFLOP/s | GB/s (alg) |", "|---|---|---|---|---|"]
    for name, k, ms, tf, gb in layer_rows(f"{ROOT}/profiles/{tag}_{which}_per_layer.txt"):
        out.append(f"| {name} | {k} | {ms:.3f} | {tf:.0f} | {gb} |")
    return "\n".join(out)


def fp32_families(tag):
    b = json.load(open(f"{ROOT}/profiles/{tag}_bench.json"))
    fam = {f["kernel"]: f for f in b["roofline"]["families"]}
    pmc = json.load(open(f"{ROOT}/profiles/{tag}_pmc_fp32.json"))["kernels"]
    per_step = {}
    for _, k, *_ in layer_rows(f"{ROOT}/profiles/{tag}_per_layer.txt"):
        per_step[k.split("+")[0]] = per_step.get(k.split("+")[0], 0) + 1
    names = {"conv3x3_wino4a": "conv3x3_wino4a_f32", "conv3x3_wino4b": "conv3x3_wino4b_f32", "conv3x3_wino4s": "miunet::conv3x3_wino4s_f32<*>",
             "convT2x2_taps": "miunet::convT2x2_taps_f32<*>", "conv3x3_wino4": "miunet::conv3x3_wino4_f32<*>", "conv3x3_first": "miunet::conv3x3_first_kernel<*>"}
    out = ["| kernel family | launches / step | share of device time | algorithmic TF/s | executed ÷ 157.3 (`roofline.frac` for the dominant one) | PMC: MFMA busy at the nominal 2.4 GHz "
           "| … at the clock the launch ran at | `SQ_WAIT_ANY` ÷ wave cycles | HBM bytes / launch (PMC) vs algorithmic |", "|---|---|---|---|---|---|---|---|---|"]
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["share_of_device_time"]):
        q = pmc.get(names.get(k, k))
        if q is None:
            continue
        alg = f["algorithmic_gbs"] * f["avg_launch_ms"] / 1e3
        out.append(f"| `{k}` | {per_step.get(k, '?')} | {f['share_of_device_time']:.3f} | {f['algorithmic_tflops']:.1f} | {f['frac_of_mfma_peak']:.3f} | {q['mfma_busy']:.3f} | "
                   f"{q['mfma_busy_at_measured_clock']:.3f} ({q['clock_ghz_from_sq_busy']:.2f} GHz) | {q['sq_wait_any_per_launch'] / q['sq_wave_cycles_per_launch'] * 100:.1f} % | "
                   f"{q['hbm_bytes_per_launch'] / 1e9:.2f} GB vs {alg:.2f} GB |")
    return "\n".join(out)


def lp_counters(tag, which):
    pmc = json.load(open(f"{ROOT}/profiles/{tag}_pmc_{which}.json"))["kernels"]
    parts = []
    for k in ("conv3x3_lp2n", "conv3x3_lpr", "conv3x3_lprk", "conv_mfma_bf16", "convT2x2_lpr"):
        q = pmc.get(f"miunet::{k}<*>")
        if q:
            parts.append(f"`{k}` {q['mfma_busy']:.2f} busy ({q['mfma_busy_at_measured_clock']:.2f} at {q['clock_ghz_from_sq_busy']:.2f} GHz), {q['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch")
    return f"Counters (`profiles/{tag}_pmc_{which}.json`): " + "; ".join(parts) + "."


def main():
    tag = sys.argv[1]
    gens = {"fp32_layers": lambda: fp32_layers(tag), "fp32_families": lambda: fp32_families(tag), "bf16_layers": lambda: lp_layers(tag, "bf16"),
            "fp16_layers": lambda: lp_layers(tag, "fp16"), "bf16_counters": lambda: lp_counters(tag, "bf16"), "fp16_counters": lambda: lp_counters(tag, "fp16")}
    path = f"{ROOT}/DESIGN.md"
    s = open(path).read()

    def sub(m):
        name = m.group(1)
        if name not in gens:
            raise SystemExit(f"unknown table {name}")
        return f"<!-- table:{name} -->\n{gens[name]()}\n<!-- /table -->"
    s, n = re.subn(r"<!-- table:(\w+) -->\n.*?\n<!-- /table -->", sub, s, flags=re.S)
    open(path, "w").write(s)
    print(f"{n} tables refreshed from profiles/{tag}_*")


if __name__ == "__main__":
    main()
