#!/usr/bin/env python3
"""Golden JSON bytes from the REFERENCE's own vendored nlohmann/json.hpp (3.12.0), via oracle/_ref/json_probe
(oracle/json_probe.cpp, built by oracle/Makefile from /root/reference/include where it lies).  The outputs are data:
tests/golden/json/<case>.json plus cases.json describing the inputs.  Run in the build container only."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "json_probe")
OUT = os.path.join(HERE, "json")

SIZE_CASES = [
    ("size_a", "a.raw", 2048, 1536, 512, 512),
    ("size_tif", "scan 01.tif", 300, 200, 512, 512),
    ("size_quote", 'we"ird\\na\tme.raw', 1, 65535, 512, 512),
    ("size_utf8", "影像_7.raw", 4096, 4096, 512, 512),
]
POLY_CASES = [
    ("poly_one", "a", 2048, 1536, [[(4, 8), (12, 8)]]),
    ("poly_rects", "case_17", 1024, 768, [[(10, 20), (10, 60), (90, 60), (90, 20)], [(0, 0)], [(5, 5), (6, 6), (7, 5)]]),
    ("poly_neg_big", "x", 70000, 3, [[(-5, 0), (2147483647, -2147483648)]]),
    ("poly_name", 'q"\\é', 512, 512, [[(1, 2), (3, 4), (5, 6)]]),
    ("poly_many", "m", 640, 480, [[(i, (i * 7) % 13) for i in range(40)]] + [[(k, k)] for k in range(5)]),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    index = {"size": [], "poly": []}
    for name, fn, w, h, sw, sh in SIZE_CASES:
        out = subprocess.run([PROBE, "size", fn, str(w), str(h), str(sw), str(sh)], check=True, capture_output=True).stdout
        open(os.path.join(OUT, name + ".json"), "wb").write(out)
        index["size"].append({"case": name, "raw_filename": fn, "w": w, "h": h, "scaled_w": sw, "scaled_h": sh})
    for name, base, ow, oh, contours in POLY_CASES:
        stdin = f"{len(contours)}\n" + "".join(f"{len(c)} " + " ".join(f"{x} {y}" for x, y in c) + "\n" for c in contours)
        out = subprocess.run([PROBE, "poly", base, str(ow), str(oh)], input=stdin.encode(), check=True, capture_output=True).stdout
        open(os.path.join(OUT, name + ".json"), "wb").write(out)
        index["poly"].append({"case": name, "base_name": base, "original_width": ow, "original_height": oh, "contours": contours})
    json.dump(index, open(os.path.join(OUT, "cases.json"), "w"), indent=1, ensure_ascii=True)
    print("wrote", len(SIZE_CASES) + len(POLY_CASES), "documents to", OUT)


if __name__ == "__main__":
    main()
