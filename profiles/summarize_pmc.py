#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into a
per-kernel HBM-traffic summary.  gfx950 corrections from that guide: FETCH_SIZE (KB) reports exactly half of a wide
coalesced read stream -> doubled; WRITE_SIZE (KB) is exact for streaming stores.  Both calibrate on this workload's
own known-byte kernels: maxpool reads 480 MiB/launch on average (counter: 240 MiB), conv3x3_first writes exactly 1 GiB.

usage: summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d


f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
# kernel families: template instantiations of one kernel are also summed into "<name>" without its <...> arguments
for d in (f, w):
    for k in list(d):
        fam = k.split("<")[0]
        if fam != k:
            d.setdefault(fam + "<*>", []).extend(d[k])
out = {}
for k in f:
    if k not in w:
        continue
    fetch = 2.0 * 1024.0 * sum(f[k]) / len(f[k])
    write = 1024.0 * sum(w[k]) / len(w[k])
    out[k] = {"launches_sampled": len(f[k]), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": fetch + write}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
