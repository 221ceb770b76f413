"""ctypes view of oracle/liboracle.so -- the CHECKER.  Imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        L = C.CDLL(so)
        L.orc_unet_forward.argtypes = [C.c_void_p, C.c_size_t, _u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_unet_forward.restype = C.c_int
        L.orc_unet_forward_bf16.argtypes = L.orc_unet_forward.argtypes
        L.orc_unet_forward_bf16.restype = C.c_int
        L.orc_unet_forward_fp16.argtypes = L.orc_unet_forward.argtypes
        L.orc_unet_forward_fp16.restype = C.c_int
        L.orc_fp16_round.argtypes = [C.c_float]
        L.orc_fp16_round.restype = C.c_float
        L.orc_bf16_round.argtypes = [C.c_float]
        L.orc_bf16_round.restype = C.c_float
        L.orc_num_threads.restype = C.c_int
        L.orc_normalize_u8.argtypes = [_u8p, C.c_size_t, _f32p]
        L.orc_argmax_planar.argtypes = [_f32p, C.c_int, C.c_size_t, _u8p]
        L.orc_conv3x3.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p]
        L.orc_bn_relu.argtypes = [_f32p, C.c_size_t, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_int]
        L.orc_maxpool2x2.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_convT2x2.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, _f32p, C.c_int, C.c_int]
        L.orc_conv1x1_planar.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, _f32p]
        L.orc_minmax_u16.argtypes = [_u16p, C.c_size_t, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.orc_preprocess_raw.argtypes = [_u16p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.orc_mask_to_image.argtypes = [_u8p, C.c_size_t, _u8p]
        L.orc_connected_components8.argtypes = [_u8p, C.c_int, C.c_int, _i32p, _i32p, C.c_int]
        L.orc_connected_components8.restype = C.c_int
        L.orc_fill_holes.argtypes = [_u8p, C.c_int, C.c_int]
        L.orc_open3x3.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        L.orc_postprocess_mask.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        _LIB = L
    return _LIB


def bf16_round(a):
    """round-to-nearest-even to bfloat16, kept in float32 (numpy mirror of the oracle's bf16_round)"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.reshape(np.shape(a))


def fp16_round(a):
    """round-to-nearest-even to IEEE binary16, kept in float32 (numpy's own conversion)"""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def unet_forward(blob: bytes, imgs: np.ndarray, want_logits=True, nthreads=0, bf16=False, fp16=False):
    """imgs u8 [B,H,W,C] -> (logits f32 [B,classes,H,W] or None, labels u8 [B,H,W]); bf16=True emulates BASELINE
    config 3 (bf16 conv operands, fp32 accumulate)"""
    L = lib()
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    b, h, w, _ = imgs.shape
    classes = int.from_bytes(blob[24:28], "little")
    logits = np.empty((b, classes, h, w), np.float32) if want_logits else None
    labels = np.empty((b, h, w), np.uint8)
    buf = C.create_string_buffer(blob, len(blob)) if not isinstance(blob, C.Array) else blob
    fn = L.orc_unet_forward_bf16 if bf16 else L.orc_unet_forward_fp16 if fp16 else L.orc_unet_forward
    rc = fn(C.cast(buf, C.c_void_p), len(blob), imgs, b, h, w,
            logits.ctypes.data if want_logits else None, labels.ctypes.data, nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_unet_forward failed rc={rc}")
    return logits, labels


def conv3x3(x, w):
    b, h, ww, cin = x.shape
    cout = w.shape[0]
    out = np.empty((b, h, ww, cout), np.float32)
    lib().orc_conv3x3(np.ascontiguousarray(x, np.float32), b, h, ww, cin, np.ascontiguousarray(w, np.float32), cout, out)
    return out


def bn_relu(x, gamma, beta, mean, var, eps=1e-5, relu=True):
    y = np.ascontiguousarray(x, np.float32).copy()
    c = y.shape[-1]
    lib().orc_bn_relu(y, y.size // c, c, *(np.ascontiguousarray(t, np.float32) for t in (gamma, beta, mean, var)), eps, int(relu))
    return y


def maxpool2x2(x):
    b, h, w, c = x.shape
    out = np.empty((b, h // 2, w // 2, c), np.float32)
    lib().orc_maxpool2x2(np.ascontiguousarray(x, np.float32), b, h, w, c, out)
    return out


def convT2x2(x, w, bias):
    b, h, ww, cin = x.shape
    cout = w.shape[1]
    out = np.empty((b, 2 * h, 2 * ww, cout), np.float32)
    lib().orc_convT2x2(np.ascontiguousarray(x, np.float32), b, h, ww, cin, np.ascontiguousarray(w, np.float32),
                       np.ascontiguousarray(bias, np.float32), cout, out, cout, 0)
    return out


def conv1x1_planar(x, w, bias):
    b, h, ww, cin = x.shape
    k = w.shape[0]
    out = np.empty((b, k, h, ww), np.float32)
    lib().orc_conv1x1_planar(np.ascontiguousarray(x, np.float32), b, h, ww, cin, np.ascontiguousarray(w, np.float32),
                             np.ascontiguousarray(bias, np.float32), k, out)
    return out


def argmax_planar(logits):
    k, h, w = logits.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_argmax_planar(np.ascontiguousarray(logits, np.float32), k, h * w, out)
    return out


def normalize_u8(a):
    a = np.ascontiguousarray(a, np.uint8)
    out = np.empty(a.shape, np.float32)
    lib().orc_normalize_u8(a.reshape(-1), a.size, out.reshape(-1))
    return out


def preprocess_raw(raw, out_w=512, out_h=512):
    raw = np.ascontiguousarray(raw, np.uint16)
    h, w = raw.shape
    out = np.empty((out_h, out_w), np.uint8)
    lib().orc_preprocess_raw(raw, w, h, out, out_w, out_h)
    return out


def postprocess_mask(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    out = np.empty_like(mask)
    lib().orc_postprocess_mask(mask, out, w, h)
    return out


def fill_holes(mask):
    m = np.ascontiguousarray(mask, np.uint8).copy()
    lib().orc_fill_holes(m, m.shape[1], m.shape[0])
    return m


def open3x3(binimg):
    b = np.ascontiguousarray(binimg, np.uint8)
    out = np.empty_like(b)
    lib().orc_open3x3(b, out, b.shape[1], b.shape[0])
    return out


def connected_components8(fg):
    fg = np.ascontiguousarray(fg, np.uint8)
    h, w = fg.shape
    labels = np.empty((h, w), np.int32)
    stats = np.zeros((h * w // 2 + 2, 5), np.int32)
    nc = lib().orc_connected_components8(fg, w, h, labels, stats, stats.shape[0])
    return nc, labels, stats[:nc]


def mask_to_image(mask):
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(m)
    lib().orc_mask_to_image(m.reshape(-1), m.size, out.reshape(-1))
    return out


def find_contours(mask):
    """list of [(x, y), ...] as extract_contours (threshold 127, external, simple)"""
    m = np.ascontiguousarray(mask, np.uint8)
    h, w = m.shape
    L = lib()
    L.orc_find_contours.argtypes = [_u8p, C.c_int, C.c_int, _i32p, C.c_int, _i32p, C.c_int]
    L.orc_find_contours.restype = C.c_int
    cap_p, cap_c = 2 * h * w + 16, h * w + 2
    xy = np.zeros((cap_p, 2), np.int32)
    st = np.zeros(cap_c + 1, np.int32)
    n = L.orc_find_contours(m, w, h, xy.reshape(-1), cap_p, st, cap_c)
    if n < 0:
        raise RuntimeError("contour capacity")
    return [[tuple(p) for p in xy[st[i]:st[i + 1]].tolist()] for i in range(n)]


def map_points(pts, sx, sy):
    a = np.ascontiguousarray(np.array(pts, np.int32).reshape(-1, 2))
    out = np.empty_like(a)
    L = lib()
    L.orc_map_points.argtypes = [_i32p, C.c_int, C.c_double, C.c_double, _i32p]
    L.orc_map_points(a.reshape(-1), a.shape[0], sx, sy, out.reshape(-1))
    return [tuple(p) for p in out.tolist()]
