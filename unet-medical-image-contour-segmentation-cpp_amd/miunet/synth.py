"""Deterministic synthetic inputs (there is no dataset and no published weights: SURVEY.md §0.1).

Everything is counter-based splitmix64 in uint64 numpy arithmetic, so a (seed, index) pair gives the
same bytes on every machine; the gaussian-like draws are a sum of four uniforms (Irwin-Hall, unit
variance) so no libm call can change a bit between this container and the GPU box.
"""
from __future__ import annotations

import numpy as np

from .spec import UNetSpec

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit draws: draw i is mix(seed + (offset+i+1)*GOLD)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 in [0,1), exact (53 random bits)."""
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normalish(seed: int, n: int) -> np.ndarray:
    """Unit-variance, zero-mean, bell-shaped float64 draws: Irwin-Hall sum of the four 16-bit fields of ONE
    64-bit draw (exact integer arithmetic; float only in the final affine map)."""
    z = splitmix64(seed, n)
    m = np.uint64(0xFFFF)
    s = (z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))
    # each field is uniform on {0..65535}: mean 32767.5, variance (65536^2 - 1) / 12
    return (s.astype(np.float64) - 131070.0) * (1.0 / np.sqrt((65536.0 ** 2 - 1.0) / 3.0))


def _seed_for(seed: int, name: str) -> int:
    h = seed & 0xFFFFFFFFFFFFFFFF
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def make_weights(spec: UNetSpec, seed: int = 1234) -> dict:
    """He-normal conv weights, BN close to identity with jitter (so folding is exercised), distinct head biases."""
    t = {}
    for name, shape in spec.tensor_list():
        n = int(np.prod(shape))
        s = _seed_for(seed, name)
        if name.endswith(".w"):
            if ".t." in name:          # convT [Cin][Cout][2][2]: each output pixel sees Cin taps
                fan_in = shape[0]
                std = np.sqrt(1.0 / fan_in)
            elif name == "outc.w":
                fan_in = shape[1]
                std = np.sqrt(2.0 / fan_in)
            else:
                fan_in = shape[1] * 9
                std = np.sqrt(2.0 / fan_in)
            v = normalish(s, n) * std
        elif name.endswith(".gamma"):
            v = 1.0 + 0.1 * (uniform01(s, n) - 0.5)
        elif name.endswith(".beta"):
            v = 0.1 * (uniform01(s, n) - 0.5)
        elif name.endswith(".mean"):
            v = 0.1 * (uniform01(s, n) - 0.5)
        elif name.endswith(".var"):
            v = 1.0 + 0.2 * (uniform01(s, n) - 0.5)
        elif name == "outc.b":
            v = np.array([0.05 * ((i * 7) % 5 - 2) for i in range(n)], dtype=np.float64)
        else:                           # convT bias
            v = 0.05 * (uniform01(s, n) - 0.5)
        t[name] = v.astype(np.float32).reshape(shape)
    return t


def make_images(b: int, h: int, w: int, c: int = 1, seed: int = 0x5EED, kind: str = "bytes") -> np.ndarray:
    """u8 [B,H,W,C].  kind="bytes": uniform random bytes (throughput; conv time is data independent).
    kind="blobs": a few soft ellipses on a ramp + noise (end-to-end runs)."""
    if kind == "bytes":
        out = np.empty((b, h, w, c), dtype=np.uint8)
        for i in range(b):
            z = splitmix64(seed + i, (h * w * c + 7) // 8)
            out[i] = z.view(np.uint8)[: h * w * c].reshape(h, w, c)
        return out
    if kind != "blobs":
        raise ValueError(kind)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.empty((b, h, w, c), dtype=np.uint8)
    for i in range(b):
        p = uniform01(seed + i, 64)
        img = 30.0 + 40.0 * xx / w
        for k in range(4):
            cx, cy = p[5 * k] * w, p[5 * k + 1] * h
            rx, ry = (0.08 + 0.25 * p[5 * k + 2]) * w, (0.08 + 0.25 * p[5 * k + 3]) * h
            d = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2
            img += (60.0 + 120.0 * p[5 * k + 4]) / (1.0 + d * d * d)
        noise = (splitmix64(seed + 7919 * (i + 1), h * w) >> np.uint64(59)).astype(np.float64).reshape(h, w)
        img = np.clip(np.floor(img + noise), 0, 255)
        out[i] = img.astype(np.uint8)[..., None].repeat(c, axis=2)
    return out


def make_raw16(h: int, w: int, seed: int = 77, lo: int = 50, hi: int = 4000) -> np.ndarray:
    """u16 [h][w] 12-bit-like detector image for the RAW path (src/preprocess.cpp:76)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    p = uniform01(seed, 16)
    cx, cy = (0.3 + 0.4 * p[0]) * w, (0.3 + 0.4 * p[1]) * h
    d = ((xx - cx) / (0.3 * w)) ** 2 + ((yy - cy) / (0.25 * h)) ** 2
    img = lo + (hi - lo) * (0.15 + 0.8 / (1.0 + d * d))
    noise = (splitmix64(seed ^ 0xABCDEF, h * w) >> np.uint64(58)).astype(np.float64).reshape(h, w)
    return np.clip(np.floor(img + noise), lo, hi).astype(np.uint16)


def make_threshold_weights(spec: UNetSpec) -> dict:
    """Structured weights for end-to-end runs: random weights give speckle label maps that postprocess_mask erases
    (nothing reaches 6 % of the image), so the polygon half would never run.  Here the network is an intensity
    classifier routed through the top skip connection: channel 0 carries x = pixel/255 unchanged (centre taps, BN
    scale 1), everything else is zero, and the head is  class0 = 0.301, class1 = 0.5 x + 0.1, class2 = x - 0.15
    =>  x < 0.402 -> 0,  0.402 < x < 0.5 -> 1,  x > 0.5 -> 2.  Both thresholds fall BETWEEN 8-bit levels (102.51 and 127.5
    of 255), so no pixel value produces tied logits: the smallest margin is ~1e-3 and label maps are comparable exactly."""
    t = {}
    for name, shape in spec.tensor_list():
        if name.endswith(".gamma"):
            v = np.ones(shape, np.float32)
        elif name.endswith(".var"):
            v = np.full(shape, 1.0 - spec.bn_eps, np.float32)
        else:
            v = np.zeros(shape, np.float32)
        t[name] = v
    L = spec.levels
    t["inc.c1.w"][0, 0, 1, 1] = 1.0
    t["inc.c2.w"][0, 0, 1, 1] = 1.0
    t[f"up{L}.c1.w"][0, 0, 1, 1] = 1.0          # input channel 0 of the concat = skip channel 0
    t[f"up{L}.c2.w"][0, 0, 1, 1] = 1.0
    t["outc.w"][1, 0] = 0.5
    t["outc.w"][2, 0] = 1.0
    t["outc.b"][:3] = [0.301, 0.1, -0.15]
    return t
