#!/usr/bin/env python3
"""Bring-up of the assembly F(4x4,3x3) kernel (conv3x3_wino4a): each case in its own child process with a deadline (a hang or a
fault costs that case, not the session), compared with the oracle and with the hipcc kernel of the same layer.
usage: asm_bringup.py [case ...]   (case = B,H,W,Cin,Cout[,pool][,exact])"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, numpy as np
sys.path[:0] = [%(pkg)r, %(tests)r]
import oracle_lib as orc
from miunet import binding
B, H, W, Cin, Cout = %(shape)s
pool, exact = %(pool)s, %(exact)s
r = np.random.default_rng(B * 1000 + H * 100 + W + Cin + Cout)
if exact:
    x = r.integers(-2, 3, (B, H, W, Cin)).astype(np.float32)
    w = np.zeros((Cout, Cin, 3, 3), np.float32)
    for (ky, kx, ci, co) in [(0, 2, 5, 7), (2, 0, Cin - 1, Cout - 1), (1, 1, 0, 0), (0, 0, 17, 33), (2, 2, 9, 40)]:
        w[co, ci, ky, kx] = 576.0
    scale = shift = None
    relu = False
else:
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    scale = (1.0 + 0.1 * r.standard_normal(Cout)).astype(np.float32)
    shift = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    relu = True
op = %(op)r + ("_pool" if pool else "")
got = binding.layer_debug(op, x, w, scale, shift, relu=relu)
ref = orc.conv3x3(x, w)
if scale is not None:
    ref = np.maximum(ref * scale + shift, 0.0)
if pool:
    ref = ref.reshape(B, H // 2, 2, W // 2, 2, Cout).max(axis=(2, 4))
nan = int(np.isnan(got).sum())
err = float(np.nanmax(np.abs(got - ref))) if nan < got.size else float("nan")
tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
bad = np.argwhere(~(np.abs(got - ref) < tol))
print(f"  max|err| {err:.3e} (tol {tol:.1e}), NaN (unwritten) {nan}, elements outside tolerance {len(bad)} of {got.size}", flush=True)
if len(bad):
    print("  first bad (b, y, x, c):", bad[:6].tolist(), "got", [float(got[tuple(i)]) for i in bad[:3]], "ref", [float(ref[tuple(i)]) for i in bad[:3]])
    ys = sorted(set(int(i[1]) for i in bad)); xs = sorted(set(int(i[2]) for i in bad)); cs = sorted(set(int(i[3]) for i in bad))
    print("  bad rows", ys[:40], "\n  bad cols", xs[:40], "\n  bad channels", cs[:48], "...", len(cs))
if exact:
    print("  bit-exact:", bool(np.array_equal(got, ref)))
sys.exit(0 if (len(bad) == 0 and nan == 0) else 1)
"""

DEFAULT = ["1,16,16,64,128", "1,16,16,64,128,exact", "1,32,32,64,128", "2,32,48,128,256", "1,32,32,64,128,pool", "8,128,128,64,128",
           "4,64,64,256,256,pool", "16,32,32,512,512"]


def main():
    cases = sys.argv[1:] or DEFAULT
    env = dict(os.environ)
    env.setdefault("MIUNET_WINO4S", "0")
    failed = 0
    for c in cases:
        parts = c.split(",")
        shape = tuple(int(p) for p in parts[:5])
        src = CHILD % {"pkg": os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"), "tests": os.path.join(ROOT, "tests"),
                       "shape": repr(shape), "pool": "pool" in parts, "exact": "exact" in parts, "op": os.environ.get("ASM_OP", "conv3x3_wino4a")}
        print(f"case {c}:", flush=True)
        try:
            r = subprocess.run([sys.executable, "-c", src], env=env, timeout=120)
            if r.returncode != 0:
                failed += 1
                print(f"  -> exit code {r.returncode}", flush=True)
                if r.returncode < 0:
                    print("  (killed by a signal: stopping, nothing further runs on this GPU)", flush=True)
                    break
        except subprocess.TimeoutExpired:
            print("  -> TIMEOUT (hang): stopping, nothing further runs on this GPU", flush=True)
            failed += 1
            break
    print(f"{failed} failing case(s)")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
