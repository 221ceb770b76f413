#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes of `bench.py` (separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes) into a
per-kernel summary that bench.py quotes in its `roofline` object:

  FETCH_SIZE / WRITE_SIZE  -> HBM bytes per launch.  gfx950 corrections from the guide: FETCH_SIZE (KB) reports exactly
                              half of a wide coalesced read stream -> doubled; WRITE_SIZE (KB) is exact for streaming
                              stores.  Both calibrate on this workload's known-byte kernels (conv3x3_first writes exactly
                              1 GiB at batch 16).
  SQ_VALU_MFMA_BUSY_CYCLES -> matrix-pipe busy fraction.  The counter adds the issue cycles of every MFMA on every SIMD
                              (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32 and per 32x32x16 bf16, 16 per 16x16x32 bf16 --
                              checked against instruction counts of kernels whose MFMA count is known), so
                              mfma_busy = cycles / (1024 SIMDs x launch duration x 2.4 GHz), the launch duration taken from
                              the same pass's dispatch timestamps -- the fraction of the NOMINAL peak, comparable with
                              bench.py's roofline.frac.  The chip runs these kernels below 2.4 GHz, so the same pass's
                              SQ_BUSY_CYCLES gives the clock (clock_ghz_from_sq_busy) and
                              mfma_busy_at_measured_clock = cycles / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines).

The output records a hash of the kernel sources it was measured on; bench.py refuses to quote a summary whose hash is not
the tree's.

usage: summarize_pmc.py --out profiles/rNN_pmc_<fp32|bf16|fp16>.json [--fetch F.csv] [--write W.csv] [--sq S.csv]"""
import argparse
import collections
import csv
import hashlib
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4
CLOCK_HZ = 2.4e9              # nominal; bench.py prices its peaks (157.3 TF fp32, 2.5 PF 16-bit dense) at this clock too
SHADER_ENGINES = 32           # 8 XCDs x 4


def kernel_source_sha():
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(pkg, "*")) + glob.glob(os.path.join(pkg, "asm", "*"))):      # the assembly generator is kernel source too
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def short(name):
    """kernel name without its argument list; names rocprofv3 left mangled (_ZN6miunet14conv_mfma_bf16I...) still get
    their namespace::function prefix so that they fall into the right family"""
    m = re.match(r"_ZN(\d+)([A-Za-z_]\w*)", name)
    if m:
        ns = m.group(2)[:int(m.group(1))]
        rest = m.group(2)[int(m.group(1)):] + name[m.end():]
        m2 = re.match(r"(\d+)", rest)
        if m2:
            n = int(m2.group(1))
            fn = rest[len(m2.group(1)):len(m2.group(1)) + n]
            return f"{ns}::{fn}<{rest[len(m2.group(1)) + n:][:40]}>"
    return name.split("(")[0].replace("void ", "").strip()


def agg(path, counters):
    """{kernel: {counter: [values]}, ...} plus per-kernel dispatch durations (ns)"""
    val = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] in counters:
            val[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return val, dur


def with_families(d):
    """template instantiations of one kernel are also summed into "<name><*>" """
    for k in list(d):
        fam = k.split("<")[0]
        if fam != k:
            tgt = d[fam + "<*>"]
            if isinstance(d[k], dict):
                for c, v in d[k].items():
                    if isinstance(v, list):
                        tgt.setdefault(c, []).extend(v)
                    else:
                        tgt[c] = v
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--sha", default=None, help="source hash of the tree the passes ran on (default: this tree's)")
    a = ap.parse_args()
    out = collections.defaultdict(dict)
    mean = lambda v: sum(v) / len(v)
    if a.fetch and a.write:
        f, _ = agg(a.fetch, {"FETCH_SIZE"})
        w, _ = agg(a.write, {"WRITE_SIZE"})
        f, w = with_families(f), with_families(w)
        for k in f:
            if k not in w or not f[k]["FETCH_SIZE"] or not w[k]["WRITE_SIZE"]:
                continue
            fetch = 2.0 * 1024.0 * mean(f[k]["FETCH_SIZE"])
            write = 1024.0 * mean(w[k]["WRITE_SIZE"])
            out[k].update({"launches_sampled": len(f[k]["FETCH_SIZE"]), "fetch_bytes_per_launch": fetch,
                           "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write})
    if a.sq:
        names = {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                 "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16",
                 "SQ_INSTS_MFMA", "SQ_VALU_MFMA_COEXEC_CYCLES"}
        s, dur = agg(a.sq, names)
        for k in list(dur):
            s[k]["__dur"] = list(dur[k].values())
        s = with_families(s)
        for k, c in s.items():
            if not c.get("SQ_VALU_MFMA_BUSY_CYCLES") or not c.get("__dur"):
                continue
            busy, d_ns = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(c["__dur"])
            rec = {"sq_launches_sampled": len(c["__dur"]), "avg_launch_ns_under_pmc": d_ns, "mfma_busy_cycles_per_launch": busy,
                   "mfma_busy": busy / (SIMDS * d_ns * 1e-9 * CLOCK_HZ)}
            if c.get("SQ_BUSY_CYCLES"):
                # the denominator at the clock the launch actually ran at: SQ_BUSY_CYCLES counts, per shader engine (32 on
                # the chip), the cycles its sequencers had work -- for a launch that fills the chip that is the launch's
                # length in shader clocks (2.27-2.30 GHz on every long kernel of r02m, not the nominal 2.4)
                sq_cycles = mean(c["SQ_BUSY_CYCLES"]) / SHADER_ENGINES
                rec["clock_ghz_from_sq_busy"] = sq_cycles / d_ns
                rec["mfma_busy_at_measured_clock"] = busy / (SIMDS * sq_cycles)
            for n in sorted(names - {"SQ_VALU_MFMA_BUSY_CYCLES"}):
                if c.get(n):
                    rec[n.lower() + "_per_launch"] = mean(c[n])
            out[k].update(rec)
    doc = {"kernel_source_sha": a.sha or kernel_source_sha(), "command": a.command,
           "method": "rocprofv3 --pmc, one counter group per run, --kernel-trace only; FETCH_SIZE doubled (gfx950), "
                     "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)",
           "kernels": out}
    json.dump(doc, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if "<*>" in k or "<" not in k}, indent=1))


if __name__ == "__main__":
    main()
