#!/usr/bin/env python3
"""dev: substitute the R03_* placeholders of DESIGN.md with numbers read from gpurun_out/<tag>/ (tools/profile_round.sh output),
so that every figure in the document comes from ONE profile run of the final tree."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
D = os.path.join(ROOT, "gpurun_out", tag)


def last_json(p):
    return json.loads(open(os.path.join(D, p)).read().strip().splitlines()[-1])


def layers(p):
    rows = []
    for l in open(os.path.join(D, p)):
        m = re.match(r"(\S+)\s+(\S+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", l)
        if m:
            rows.append((m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))))
    return rows


b = last_json("bench.json")
r = b["roofline"]
pm = json.load(open(os.path.join(D, "pmc_fp32.json")))["kernels"]
names = {"conv3x3_wino4": "miunet::conv3x3_wino4_f32<*>", "conv3x3_wino4s": "miunet::conv3x3_wino4s_f32<*>",
         "convT2x2_taps": "miunet::convT2x2_taps_f32<*>", "conv3x3_first": "miunet::conv3x3_first_kernel<*>"}
fam = {f["kernel"]: f for f in r["families"]}
L = layers("per_layer.txt")

out = {}
t = ["| layer | kernel | ms / launch | algorithmic TFLOP/s | executed ÷ 157.3 TF/s | algorithmic GB/s |", "|---|---|---|---|---|---|"]
for n, k, ms, tf, gb in L:
    red = 4.0 if "wino4" in k else 1.0
    t.append(f"| {n} | {k} | {ms:.3f} | {tf:.1f} | {tf / red / 157.3:.2f} | {gb} |")
t.append("")
t.append(f"Step {b['ms_per_step']:.2f} ms = **{b['value']:.0f} images/s** on hipGraph replay ({sum(x[2] for x in L):.2f} ms of kernels in the eager profiled pass); "
         f"whole network {r['whole_net_algorithmic_tflops']:.0f} algorithmic TFLOP/s.  By kernel family (`profiles/{tag}_bench.json`, counters `profiles/{tag}_pmc_fp32.json`):")
t.append("")
t.append("| kernel family | launches / step | share of device time | algorithmic TF/s | executed ÷ 157.3 (`roofline.frac` for the dominant one) | PMC: MFMA busy at the nominal 2.4 GHz | … at the clock the launch ran at | HBM bytes / launch (PMC) vs algorithmic |")
t.append("|---|---|---|---|---|---|---|---|")
for f in r["families"]:
    c = pm.get(names.get(f["kernel"], ""), {})
    alg_bytes = f["algorithmic_gbs"] * f["avg_launch_ms"] / 1e3
    t.append(f"| `{f['kernel']}` | {f['launches'] // b['steps']} | {f['share_of_device_time']:.3f} | {f['algorithmic_tflops']:.1f} | {f['frac_of_mfma_peak']:.3f} | "
             f"{c.get('mfma_busy', 0):.3f} | {c.get('mfma_busy_at_measured_clock', 0):.3f} ({c.get('clock_ghz_from_sq_busy', 0):.2f} GHz) | {c.get('hbm_bytes_per_launch', 0) / 1e9:.2f} GB vs {alg_bytes:.2f} GB |")
out["R03_FP32_TABLE"] = "\n".join(t)


def span(kern):
    v = [tf / 4.0 / 157.3 for n, k, ms, tf, gb in L if k.split("+")[0] == kern]
    return min(v), max(v)


c = pm[names["conv3x3_wino4"]]
lo, hi = span("conv3x3_wino4")
out["R03_TWO_BLOCK_FRAC"] = (f"{lo:.2f}–{hi:.2f} of the fp32 MFMA peak executed by layer, **{fam['conv3x3_wino4']['frac_of_mfma_peak']:.3f}** over its 13 launches "
                             f"({fam['conv3x3_wino4']['algorithmic_tflops']:.0f} TF/s algorithmic); counter: {c['mfma_busy']:.3f} busy at the nominal clock, "
                             f"{c['mfma_busy_at_measured_clock']:.3f} at the {c['clock_ghz_from_sq_busy']:.2f} GHz the launches ran at; {c['hbm_bytes_per_launch'] / 1e9:.2f} GB of HBM traffic per launch "
                             f"against ≈ 0.6 GB algorithmic (U re-fetched by each XCD's L2; at ≈ 1.4 TB/s nowhere near the HBM roof)")
c = pm[names["conv3x3_wino4s"]]
lo, hi = span("conv3x3_wino4s")
out["R03_STAGED_FRAC"] = (f"{lo:.2f}–{hi:.2f} executed by layer, {fam['conv3x3_wino4s']['frac_of_mfma_peak']:.3f} over its four launches; counter: {c['mfma_busy']:.3f} busy "
                          f"({c['mfma_busy_at_measured_clock']:.3f} at the measured {c['clock_ghz_from_sq_busy']:.2f} GHz); {c['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch by counter (1.17 × the algorithmic bytes at batch 16 in round 2: the 18×18 halo of a 16×16 block)")

cfg = {("bf16" if "bf16" in x["config"] else "fp16"): x for x in b["configs"] if "parity" in x}
p = b["pipeline"]
g512 = last_json("global_batch512_bench.json")
dist = last_json("dist_rehearsal_1rank.json")
bf = last_json("bf16_bench.json")
fp = last_json("fp16_bench.json")
cb = b["cpu_baseline"]
rows = ["| BASELINE config | GPUs | dtype | batch | images/s | ms/image | whole-net algorithmic TFLOP/s | dominant kernel: executed ÷ dtype peak (counter) | parity record of the same run |",
        "|---|---|---|---|---|---|---|---|---|"]
rows.append(f"| configs[0] (CPU) | 0 | fp32 | 1 | {cb['value']:.2f} | {cb['ms_per_image']:.0f} | — | — | the oracle itself ({cb['cores']} host threads, kind `port`) |")
rows.append(f"| **configs[1]** (the metric) | 1 | fp32 | 16 | **{b['value']:.0f}** ({b['e2e_host']['value']:.0f} from host buffers) | {b['ms_per_image']:.3f} | {r['whole_net_algorithmic_tflops']:.0f} | "
            f"{r['frac']:.3f} of 157.3 TF/s ({pm[names['conv3x3_wino4']]['mfma_busy']:.2f} busy) | logits ≤ {b['parity']['max_abs_logit_err']:.1e}, {b['parity']['mismatches_above_margin']} label mismatches above the 1e-3 margin |")
for key, x, bb, pmc in (("bf16", cfg.get("bf16"), bf, "pmc_bf16.json"), ("fp16", cfg.get("fp16"), fp, "pmc_fp16.json")):
    if not x:
        continue
    pk = json.load(open(os.path.join(D, pmc)))["kernels"].get("miunet::conv3x3_lp2n<*>", {})
    name = "configs[2]" if key == "bf16" else "configs[4] (network half; the one-call pipeline half: §7)"
    batch = "128 (micro-batches of 16)" if key == "bf16" else "8 × 1024²×3"
    rows.append(f"| {name} | 1 | {key} operands / fp32 acc | {batch} | {x['value']:.0f} (profile run: {bb['value']:.0f}) | {x['ms_per_image']:.3f} | {x['roofline']['whole_net_algorithmic_tflops']:.0f} | "
                f"`conv3x3_lp2n` {x['roofline']['frac']:.3f} of 2.5 PF ({pk.get('mfma_busy', 0):.2f} busy; {pk.get('mfma_busy_at_measured_clock', 0):.2f} at the {pk.get('clock_ghz_from_sq_busy', 0):.2f} GHz it ran at) | "
                f"logits {x['parity']['max_abs_logit_err']:.3f} from the fp32 oracle ({x['parity']['max_abs_err_vs_16bit_oracle']:.3f} from the {key}-operand oracle), 0 label mismatches above the {x['parity']['margin']} margin; per launch: §2 |")
c4 = (dist.get("configs") or [{}])[0]
rows.append(f"| configs[3] | 1 of 8 (no node) | fp32 | 512 | {g512['value']:.0f} on one GPU (micro-batches of 16); one-rank RCCL rehearsal of the strong loop: {c4.get('value', 0):.0f} | {1e3 / g512['value']:.3f} | — | as configs[1] | gathered label maps verified: {c4.get('gathered_label_maps_verified')}; **N > 1 unmeasured** |")
out["R03_RESULTS_TABLE"] = "\n".join(rows)

LB = layers("bf16_per_layer.txt")
LF = layers("fp16_per_layer.txt")


def lp_table(rows_, title):
    t_ = [title, "", "| layer | kernel | ms / launch | TFLOP/s | GB/s (alg) |", "|---|---|---|---|---|"]
    for n, k, ms, tf, gb in rows_:
        t_.append(f"| {n} | {k} | {ms:.3f} | {tf:.0f} | {gb} |")
    return "\n".join(t_)


def pmc_line(f):
    pk = json.load(open(os.path.join(D, f)))["kernels"]
    parts = []
    for k in ("miunet::conv3x3_lp2n<*>", "miunet::conv3x3_lpr<*>", "miunet::conv3x3_lprk<*>", "miunet::conv_mfma_bf16<*>", "miunet::convT2x2_lpr<*>"):
        v = pk.get(k)
        if v:
            parts.append(f"`{k.split('::')[1][:-3]}` {v['mfma_busy']:.2f} busy ({v.get('mfma_busy_at_measured_clock', 0):.2f} at {v.get('clock_ghz_from_sq_busy', 0):.2f} GHz), {v.get('hbm_bytes_per_launch', 0) / 1e6:.0f} MB per launch")
    return "; ".join(parts)


out["R03_LP_TABLES"] = (lp_table(LB, f"Config 3, batch 16 per launch (`profiles/{tag}_bf16_per_layer.txt`; {bf['value']:.0f} images/s at batch 128, {sum(x[2] for x in LB):.2f} ms of kernels per micro-batch):")
                        + f"\n\nCounters (`profiles/{tag}_pmc_bf16.json`): " + pmc_line("pmc_bf16.json") + ".\n\n"
                        + lp_table(LF, f"Config 5, batch 8 (`profiles/{tag}_fp16_per_layer.txt`; {fp['value']:.0f} images/s, {sum(x[2] for x in LF):.2f} ms of kernels per step):")
                        + f"\n\nCounters (`profiles/{tag}_pmc_fp16.json`): " + pmc_line("pmc_fp16.json") + ".")
kb = [x for x in LB if x[1].endswith("16k")][0]
kf = [x for x in LF if x[1].endswith("16k")][0]
out["R03_LPRK_RESULT"] = (f"This profile run: config 3 `up4.c1` {kb[2]:.3f} ms = {kb[3]:.0f} TFLOP/s, {kb[4] / 1e3:.1f} TB/s algorithmic (round 2 on the 2×2 kernel: 0.668 ms; "
                          f"on 32×32×16: 0.631); config 5's 128 → 64 layer {kf[2]:.3f} ms.")
out["R03_C3"] = f"{cfg['bf16']['value']:.0f}"
out["R03_C3TF"] = f"{cfg['bf16']['roofline']['whole_net_algorithmic_tflops']:.0f}"
out["R03_C5"] = f"{cfg['fp16']['value']:.0f}"
out["R03_DIST"] = (f"{dist['value']:.0f} images/s weak at one rank with the RCCL gather in the timed loop; configs[3] strong loop {c4.get('value', 0):.0f} images/s at one rank, "
                   f"gathered label maps verified; `group`: {b['group']['runs'][0]['host_gather']['images_per_s']:.0f} images/s from host buffers in one process, "
                   f"{b['group']['runs'][1]['host_gather']['images_per_s']:.0f} with two ranks sharing the card")
d1, d2, fs, fd, fh, cc = (p[k] for k in ("device_one_call", "device_one_call_pinned", "facade_single_image", "facade_device", "facade_host", "cpu_chain"))
st = d1["stages_ms"]
sm = fs.get("stages_ms_mean", {})
out["R03_PIPELINE"] = "\n".join([
    "| 16 RAW16 images 2048×1536 → tiles, masks, contours (`bench.py` `pipeline`, this profile run) | images/s | ms/image |", "|---|---|---|",
    f"| `mi_unet_segment_raw16`, pageable sources (round 2: 672) | **{d1['images_per_s']:.0f}** | {d1['ms_per_image']:.2f} |",
    f"| the same from page-locked sources | {d2['images_per_s']:.0f} (same-card sweep, clean process: 795 pageable / 801 pinned, `profiles/r03_raw_pipeline_split_sweep.txt`) | {d2['ms_per_image']:.2f} |",
    f"| facade `process_image_batch`, files in, five artefacts per image out, one chunk of 16 | {fd['images_per_s']:.0f} | {fd['ms_per_image']:.2f} |",
    f"| facade `process_single_image` per file (round 2: 10.6 ms) | {fs['images_per_s']:.0f} | **{fs['single_image_ms']:.2f}** |",
    f"| facade host route (`MEDSEG_HOST_*=1`: the reference's own stage order, CPU pre/post/contours around the GPU network) | {fh['images_per_s']:.0f} | {fh['ms_per_image']:.1f} |",
    f"| the oracle's all-CPU chain, one image, {cc['cores']} threads | {cc['images_per_s']:.2f} | {cc['ms_per_image']:.0f} |", "",
    f"Device time per stage of the one-call route (`mi_unet_last_stage_ms`, 16 images): upload + preprocess {st['upload_preprocess']:.2f} ms (hidden under the network except for the first four images), "
    f"network {st['network']:.2f} ms (4 + 12 images: 0.8 ms more than 16 at once, the price of starting early), postprocess {st['postprocess']:.2f}, contours {st['contours']:.2f} "
    f"(both beside the next network), download {st['download']:.2f}.  Parity of the record: tile, mask and contours of image 0 equal the oracle chain's, pinned route equal to the pageable one.",
])
out["R03_SINGLE"] = (f"Mean over 16 files of 2048×1536: **{fs['single_image_ms']:.2f} ms per image** = device call {sm.get('device_call', 0):.2f} (upload + preprocess {sm.get('upload_preprocess', 0):.2f}, network {sm.get('network', 0):.2f}, "
                     f"postprocess {sm.get('postprocess', 0):.2f}, contours {sm.get('contours', 0):.2f}, download {sm.get('download', 0):.2f}) + artefacts {sm.get('artefacts', 0):.2f} "
                     f"(normalized.png + sizes.json {sm.get('normalized_png_and_sizes_json', 0):.2f} ‖ mask.png {sm.get('mask_png', 0):.2f} ‖ overlay.png + polygon.json {sm.get('overlay_png_and_polygon_json', 0):.2f}); "
                     "3.7–4.3 ms by run (the overlay PNG's sixteen deflate threads are the noisy term).")
out["R03_SINGLE_SHORT"] = f"10.6 → {fs['single_image_ms']:.1f} ms per 2048×1536 file"
out["R03_STEP"] = f"{b['ms_per_step']:.2f}"
out["R03_FP32"] = f"{b['value']:.0f}"
out["R03_SEG"] = f"{d1['images_per_s']:.0f} images/s against the network's {b['value']:.0f}"

p_ = os.path.join(ROOT, "DESIGN.md")
s = open(sys.argv[2] if len(sys.argv) > 2 else p_).read()          # optional: a template that still carries the placeholders
for k in sorted(out, key=len, reverse=True):
    if k not in s:
        print("placeholder missing:", k)
    s = s.replace(k, out[k])
left = re.findall(r"R03_[A-Z_]+", s)
print("left:", left)
open(p_, "w").write(s)
