#!/usr/bin/env python3
"""dev: markdown tables for DESIGN.md from the files tools/profile_round.sh wrote under gpurun_out/<tag>/"""
import json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
D = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", tag)
def last_json(p):
    return json.loads(open(os.path.join(D, p)).read().strip().splitlines()[-1])
def layers(p):
    rows = []
    for l in open(os.path.join(D, p)):
        m = re.match(r"(\S+)\s+(\S+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", l)
        if m: rows.append((m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))))
    return rows
b = last_json("bench.json")
pm = json.load(open(os.path.join(D, "pmc_fp32.json")))["kernels"]
print("### fp32 per layer (batch 16)\n")
print("| layer | kernel | ms / launch | algorithmic TFLOP/s | executed, of 157.3 TF/s | algorithmic GB/s |")
print("|---|---|---|---|---|---|")
for n, k, ms, tf, gb in layers("per_layer.txt"):
    red = 4.0 if "wino4" in k else 1.0
    print(f"| {n} | {k} | {ms:.3f} | {tf:.1f} | {tf / red / 157.3:.2f} | {gb} |")
r = b["roofline"]
print(f"\nstep {b['ms_per_step']:.3f} ms = {b['value']:.1f} images/s; dominant {r['kernel']}: frac {r['frac']:.3f}, algorithmic {r['algorithmic_tflops']:.1f} TF/s, avg launch {r['avg_launch_ms']:.3f} ms, {r['avg_launch_gflop']:.1f} GFLOP; whole net {r['whole_net_algorithmic_tflops']:.1f} TF/s")
print("\n| kernel family | launches / step | share of device time | algorithmic TF/s | executed / 157.3 | PMC: MFMA busy (nominal 2.4 GHz) | busy at the measured clock | clock GHz | HBM bytes / launch (PMC) |")
print("|---|---|---|---|---|---|---|---|---|")
names = {"conv3x3_wino4": "miunet::conv3x3_wino4_f32<*>", "conv3x3_wino4s": "miunet::conv3x3_wino4s_f32<*>", "convT2x2_taps": "miunet::convT2x2_taps_f32<*>", "conv3x3_first": "miunet::conv3x3_first_kernel<*>"}
for f in r["families"]:
    c = pm.get(names.get(f["kernel"], ""), {})
    print(f"| {f['kernel']} | {f['launches'] // b['steps']} | {f['share_of_device_time']:.3f} | {f['algorithmic_tflops']:.1f} | {f['frac_of_mfma_peak']:.3f} | {c.get('mfma_busy', 0):.3f} | {c.get('mfma_busy_at_measured_clock', 0):.3f} | {c.get('clock_ghz_from_sq_busy', 0):.2f} | {c.get('hbm_bytes_per_launch', 0) / 1e9:.2f} GB |")
for name, f in (("bf16 (config 3, batch 128 in micro-batches of 16)", "bf16"), ("fp16 (config 5, 1024x1024x3, batch 8)", "fp16")):
    bb = last_json(f + "_bench.json")
    print(f"\n### {name}: {bb['value']:.1f} images/s, {bb['ms_per_step']:.3f} ms per step, dominant {bb['roofline']['kernel']} frac {bb['roofline']['frac']:.3f}, whole net {bb['roofline']['whole_net_algorithmic_tflops']:.0f} TF/s\n")
    print("| layer | kernel | ms / launch | TFLOP/s | GB/s (alg) |")
    print("|---|---|---|---|---|")
    for n, k, ms, tf, gb in layers(f + "_per_layer.txt"):
        print(f"| {n} | {k} | {ms:.3f} | {tf:.1f} | {gb} |")
    pk = json.load(open(os.path.join(D, f"pmc_{f}.json")))["kernels"]
    for k, v in pk.items():
        if "<*>" in k and v.get("mfma_busy"):
            print(f"  PMC {k}: busy {v['mfma_busy']:.3f} (at measured clock {v.get('mfma_busy_at_measured_clock', 0):.3f}, {v.get('clock_ghz_from_sq_busy', 0):.2f} GHz), {v.get('hbm_bytes_per_launch', 0) / 1e6:.0f} MB / launch")
print("\n### records")
for c in b["configs"]:
    print(" ", c["config"][:60], round(c["value"], 1), "images/s, frac", round(c["roofline"]["frac"], 3), "parity", c["parity"])
print("  e2e_host", round(b["e2e_host"]["value"], 1), "cpu_baseline", b["cpu_baseline"])
p = b["pipeline"]
for k in ("device_one_call", "device_one_call_pinned", "facade_single_image", "facade_device", "facade_host", "cpu_chain"):
    print(" ", k, json.dumps(p[k])[:500])
print("  group", json.dumps(b["group"])[:900])
g = last_json("global_batch512_bench.json"); print("  global batch 512, 1 GPU:", round(g["value"], 1), "images/s", g["ms_per_step"])
dd = last_json("dist_rehearsal_1rank.json"); print("  dist rehearsal:", round(dd["value"], 1), json.dumps(dd.get("configs"))[:400])
