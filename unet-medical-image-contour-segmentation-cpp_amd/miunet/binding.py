"""ctypes binding of libmiunet.so (include/mi_unet.h) for the tests and bench.py.

The library is the product; this file is plumbing.  It raises if the shared object is missing or if a call fails --
there is no Python/CPU fallback of any kind.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MIUNET_LIB: another build of the same library (same-card A/B measurements of a kernel change, tools/dev/)
LIB_PATH = os.environ.get("MIUNET_LIB") or os.path.join(PKG_DIR, "libmiunet.so")

EXPORTS = [
    "mi_unet_default_config", "mi_unet_create", "mi_unet_load_weights", "mi_unet_load_weights_from_memory",
    "mi_unet_infer_u8", "mi_unet_infer_u8_device", "mi_unet_infer_raw16", "mi_unet_set_postprocess", "mi_unet_postprocess_masks", "mi_unet_extract_contours", "mi_unet_segment_raw16", "mi_unet_set_stream", "mi_unet_sync", "mi_unet_timer_begin",
    "mi_unet_timer_end", "mi_unet_set_profiling", "mi_unet_get_kernel_stats", "mi_unet_layer_debug", "mi_unet_destroy",
    "mi_unet_last_error", "mi_unet_device_count", "mi_unet_clone",
    "mi_unet_debug_layer_count", "mi_unet_debug_layer_info", "mi_unet_debug_capture", "mi_unet_last_stage_ms", "mi_unet_numeric_guard", "mi_unet_host_alloc", "mi_unet_host_free",
    "mi_unet_group_create", "mi_unet_group_clone", "mi_unet_group_size", "mi_unet_group_handle", "mi_unet_group_load_weights",
    "mi_unet_group_load_weights_from_memory", "mi_unet_group_set_gather", "mi_unet_group_set_postprocess",
    "mi_unet_group_weight_transport", "mi_unet_group_gather", "mi_unet_group_infer_u8", "mi_unet_group_infer_raw16",
    "mi_unet_group_segment_raw16", "mi_unet_group_destroy", "mi_unet_shard_range",
]


class Config(C.Structure):
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("in_ch", C.c_int), ("base", C.c_int), ("levels", C.c_int),
                ("classes", C.c_int), ("max_batch", C.c_int), ("device", C.c_int), ("conv_algo", C.c_int)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kernel", C.c_char * 32), ("flops", C.c_double), ("bytes", C.c_double),
                ("ms", C.c_float)]


class LayerInfo(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kernel", C.c_char * 32), ("kind", C.c_int), ("in_h", C.c_int), ("in_w", C.c_int),
                ("in_c", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int), ("out_c", C.c_int), ("in_bits", C.c_int),
                ("out_bits", C.c_int), ("pooled", C.c_int), ("fused_head", C.c_int), ("skipped", C.c_int), ("fused_first", C.c_int)]

    KINDS = ("first", "conv3x3", "convT2x2", "maxpool", "head")

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["name"], d["kernel"], d["kind"] = self.name.decode(), self.kernel.decode(), self.KINDS[self.kind]
        return d


class MiUnetError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmiunet error {code}: {msg}")
        self.code = code


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(make -C unet-medical-image-contour-segmentation-cpp_amd)")
        L = C.CDLL(LIB_PATH)
        L.mi_unet_last_error.restype = C.c_char_p
        L.mi_unet_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.mi_unet_load_weights.argtypes = [C.c_void_p, C.c_char_p]
        L.mi_unet_load_weights_from_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.mi_unet_infer_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mi_unet_infer_u8_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mi_unet_infer_raw16.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi_unet_set_postprocess.argtypes = [C.c_void_p, C.c_int]
        L.mi_unet_postprocess_masks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mi_unet_extract_contours.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.mi_unet_segment_raw16.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.mi_unet_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_unet_sync.argtypes = [C.c_void_p]
        L.mi_unet_timer_begin.argtypes = [C.c_void_p]
        L.mi_unet_timer_end.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.mi_unet_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.mi_unet_get_kernel_stats.argtypes = [C.c_void_p, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int)]
        L.mi_unet_layer_debug.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mi_unet_destroy.argtypes = [C.c_void_p]
        L.mi_unet_destroy.restype = None
        L.mi_unet_default_config.argtypes = [C.POINTER(Config)]
        L.mi_unet_default_config.restype = None
        L.mi_unet_clone.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.mi_unet_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.mi_unet_host_free.argtypes = [C.c_void_p]
        L.mi_unet_host_free.restype = None
        L.mi_unet_numeric_guard.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.mi_unet_numeric_guard.restype = C.c_char_p
        L.mi_unet_last_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.mi_unet_debug_layer_count.argtypes = [C.c_void_p]
        L.mi_unet_debug_layer_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(LayerInfo)]
        L.mi_unet_debug_capture.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.POINTER(LayerInfo)]
        L.mi_unet_group_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.mi_unet_group_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mi_unet_group_size.argtypes = [C.c_void_p]
        L.mi_unet_group_handle.argtypes = [C.c_void_p, C.c_int]
        L.mi_unet_group_handle.restype = C.c_void_p
        L.mi_unet_group_load_weights.argtypes = [C.c_void_p, C.c_char_p]
        L.mi_unet_group_load_weights_from_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.mi_unet_group_set_gather.argtypes = [C.c_void_p, C.c_int]
        L.mi_unet_group_set_postprocess.argtypes = [C.c_void_p, C.c_int]
        L.mi_unet_group_weight_transport.argtypes = [C.c_void_p]
        L.mi_unet_group_weight_transport.restype = C.c_char_p
        L.mi_unet_group_gather.argtypes = [C.c_void_p]
        L.mi_unet_group_infer_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mi_unet_group_infer_raw16.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi_unet_group_segment_raw16.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.mi_unet_group_destroy.argtypes = [C.c_void_p]
        L.mi_unet_group_destroy.restype = None
        L.mi_unet_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise MiUnetError(rc, lib().mi_unet_last_error().decode(errors="replace"))


class PinnedArray:
    """A numpy view of page-locked host memory (mi_unet_host_alloc): RAW images kept in one are uploaded without a staging copy.
    Keep the object alive as long as the view `.a` is used; close() (or garbage collection) releases the memory."""

    def __init__(self, shape, dtype):
        self._p = C.c_void_p()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        _check(lib().mi_unet_host_alloc(n, C.byref(self._p)))
        self.a = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(n,)).view(dtype).reshape(shape)

    def close(self):
        if self._p:
            self.a = None
            lib().mi_unet_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count() -> int:
    return int(lib().mi_unet_device_count())


def default_config() -> Config:
    c = Config()
    lib().mi_unet_default_config(C.byref(c))
    return c


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One engine handle = one GPU's context (the reference's thread-local TensorRTContext, include/process.h:13-23)."""

    CONV_ALGOS = {"auto": 0, "direct": 1, "winograd": 2, "winograd16": 3, "bf16": 4, "fp16": 5}

    def __init__(self, height=512, width=512, in_ch=1, base=64, levels=4, classes=3, max_batch=16, device=0,
                 conv_algo="auto"):
        self.cfg = Config(height, width, in_ch, base, levels, classes, max_batch, device, self.CONV_ALGOS[conv_algo])
        self._h = C.c_void_p()
        _check(lib().mi_unet_create(C.byref(self.cfg), C.byref(self._h)))

    def clone(self, max_batch=0) -> "Engine":
        """A second context on the same device sharing this engine's weight blob (mi_unet_clone)."""
        other = object.__new__(Engine)
        other.cfg = Config(self.cfg.height, self.cfg.width, self.cfg.in_ch, self.cfg.base, self.cfg.levels, self.cfg.classes,
                           max_batch if max_batch > 0 else self.cfg.max_batch, self.cfg.device, self.cfg.conv_algo)
        other._h = C.c_void_p()
        _check(lib().mi_unet_clone(self._h, max_batch, C.byref(other._h)))
        return other

    def close(self):
        if self._h:
            lib().mi_unet_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, path_or_blob):
        if isinstance(path_or_blob, (bytes, bytearray, memoryview)):
            buf = (C.c_char * len(path_or_blob)).from_buffer_copy(path_or_blob)
            _check(lib().mi_unet_load_weights_from_memory(self._h, C.cast(buf, C.c_void_p), len(path_or_blob)))
        else:
            _check(lib().mi_unet_load_weights(self._h, os.fsencode(path_or_blob)))

    def infer(self, imgs: np.ndarray, want_logits=False):
        """imgs u8 [B,H,W,C] (host) -> labels u8 [B,H,W], logits f32 [B,classes,H,W] or None"""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        b = imgs.shape[0]
        c = self.cfg
        if imgs.shape[1:] != (c.height, c.width, c.in_ch):
            raise ValueError(f"Input size must be {c.height}x{c.width}x{c.in_ch}, got {imgs.shape[1:]}")
        labels = np.empty((b, c.height, c.width), np.uint8)
        logits = np.empty((b, c.classes, c.height, c.width), np.float32) if want_logits else None
        _check(lib().mi_unet_infer_u8(self._h, _ptr(imgs), b, _ptr(labels), _ptr(logits)))
        return labels, logits

    def _raw_args(self, raws):
        """list of u16 planes (in_ch per image, image-major) -> (keep-alive list, pointer / width / height arrays, B)"""
        raws = [np.ascontiguousarray(r, dtype=np.uint16) for r in raws]
        n, c = len(raws), self.cfg
        if n % c.in_ch:
            raise ValueError(f"{n} planes for an engine with in_ch = {c.in_ch}")
        ptrs = (C.c_void_p * n)(*[r.ctypes.data for r in raws])
        ws = (C.c_int * n)(*[r.shape[1] for r in raws])
        hs = (C.c_int * n)(*[r.shape[0] for r in raws])
        return raws, ptrs, ws, hs, n // c.in_ch

    def _tile_buf(self, b):
        c = self.cfg
        return np.empty((b, c.height, c.width) if c.in_ch == 1 else (b, c.height, c.width, c.in_ch), np.uint8)

    def infer_raw16(self, raws, want_tiles=True, want_logits=False):
        """raws: list of u16 [h_i][w_i] planes, in_ch per image -> (tiles u8 [B,H,W(,C)] or None, labels u8 [B,H,W], logits or None)"""
        c = self.cfg
        raws, ptrs, ws, hs, b = self._raw_args(raws)
        tiles = self._tile_buf(b) if want_tiles else None
        labels = np.empty((b, c.height, c.width), np.uint8)
        logits = np.empty((b, c.classes, c.height, c.width), np.float32) if want_logits else None
        _check(lib().mi_unet_infer_raw16(self._h, ptrs, ws, hs, b, _ptr(tiles), _ptr(labels), _ptr(logits)))
        return tiles, labels, logits

    def set_postprocess(self, on: bool):
        _check(lib().mi_unet_set_postprocess(self._h, int(on)))

    def postprocess_masks(self, labels: np.ndarray):
        labels = np.ascontiguousarray(labels, np.uint8)
        out = np.empty_like(labels)
        _check(lib().mi_unet_postprocess_masks(self._h, _ptr(labels), labels.shape[0], _ptr(out)))
        return out

    def extract_contours(self, masks: np.ndarray, cap_points=8192, cap_contours=256):
        """masks u8 [B,H,W] -> per image a list of contours [(x, y), ...] (None where a capacity overflowed)"""
        masks = np.ascontiguousarray(masks, np.uint8)
        b = masks.shape[0]
        xy = np.zeros((b, cap_points, 2), np.int32)
        start = np.zeros((b, cap_contours + 1), np.int32)
        counts = np.zeros(b, np.int32)
        _check(lib().mi_unet_extract_contours(self._h, _ptr(masks), b, _ptr(xy), cap_points, _ptr(start), cap_contours, _ptr(counts)))
        out = []
        for i in range(b):
            if counts[i] < 0:
                out.append(None)
                continue
            out.append([[tuple(p) for p in xy[i, start[i, c]:start[i, c + 1]].tolist()] for c in range(counts[i])])
        return out

    def segment_raw16_prepare(self, raws, cap_points=8192, cap_contours=64):
        """argument block of mi_unet_segment_raw16 for these images: pointer arrays and caller-owned output buffers, built once"""
        c = self.cfg
        raws, ptrs, ws, hs, b = self._raw_args(raws)
        return dict(raws=raws, ptrs=ptrs, ws=ws, hs=hs, b=b, cap_points=cap_points, cap_contours=cap_contours,
                    tiles=self._tile_buf(b), masks=np.empty((b, c.height, c.width), np.uint8),
                    xy=np.zeros((b, cap_points, 2), np.int32), start=np.zeros((b, cap_contours + 1), np.int32),
                    counts=np.zeros(b, np.int32))

    def segment_raw16_run(self, p):
        """the C call alone (what a C or C++ host pays)"""
        _check(lib().mi_unet_segment_raw16(self._h, p["ptrs"], p["ws"], p["hs"], p["b"], _ptr(p["tiles"]), _ptr(p["masks"]), _ptr(p["xy"]),
                                           p["cap_points"], _ptr(p["start"]), p["cap_contours"], _ptr(p["counts"])))

    @staticmethod
    def segment_raw16_decode(p):
        xy, start, counts = p["xy"], p["start"], p["counts"]
        cont = []
        for i in range(p["b"]):
            cont.append(None if counts[i] < 0 else
                        [[tuple(q) for q in xy[i, start[i, k]:start[i, k + 1]].tolist()] for k in range(counts[i])])
        return p["tiles"], p["masks"], cont

    def segment_raw16(self, raws, cap_points=8192, cap_contours=64):
        """RAW16 images -> (tiles, mask images 0/255, contours per image) with every stage on the device"""
        p = self.segment_raw16_prepare(raws, cap_points, cap_contours)
        self.segment_raw16_run(p)
        return self.segment_raw16_decode(p)

    def infer_device(self, d_imgs_ptr: int, b: int, d_labels_ptr: int, d_logits_ptr: int = 0):
        _check(lib().mi_unet_infer_u8_device(self._h, C.c_void_p(d_imgs_ptr), b, C.c_void_p(d_labels_ptr),
                                             C.c_void_p(d_logits_ptr) if d_logits_ptr else None))

    def set_stream(self, stream_ptr: int, reset: bool = False):
        """Run on the caller's hipStream_t.  Handle 0 is the legacy default stream, which mi_unet_set_stream reads as
        "restore the engine's own (non-blocking) stream": work on it is NOT ordered against the default stream, so a
        zero handle is only accepted with reset=True (the caller says that is what they mean)."""
        if not stream_ptr and not reset:
            raise ValueError("stream handle 0 = restore the engine's own stream; pass reset=True or a non-default stream "
                             "(e.g. torch.cuda.Stream(dev).cuda_stream)")
        _check(lib().mi_unet_set_stream(self._h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def sync(self):
        _check(lib().mi_unet_sync(self._h))

    def timer_begin(self):
        _check(lib().mi_unet_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        _check(lib().mi_unet_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    def numeric_guard(self):
        """(text, tripped, diff) of mi_unet_numeric_guard: did this weight set keep F(4x4,3x3)?"""
        t, d = C.c_int(), C.c_float()
        text = lib().mi_unet_numeric_guard(self._h, C.byref(t), C.byref(d)).decode()
        return text, bool(t.value), float(d.value)

    STAGES = ("upload_preprocess", "network", "postprocess", "contours", "download")

    def last_stage_ms(self):
        """device time per stage of the last infer_raw16 / segment_raw16 call (mi_unet_last_stage_ms)"""
        ms = (C.c_float * len(self.STAGES))()
        _check(lib().mi_unet_last_stage_ms(self._h, ms))
        return dict(zip(self.STAGES, (float(v) for v in ms)))

    def layers(self):
        """the launch plan, step by step (mi_unet_debug_layer_info): list of dicts with name / kind / per-image shapes"""
        out = []
        for i in range(lib().mi_unet_debug_layer_count(self._h)):
            info = LayerInfo()
            _check(lib().mi_unet_debug_layer_info(self._h, i, C.byref(info)))
            out.append(info.as_dict())
        return out

    def capture(self, imgs: np.ndarray, layer: int, img: int = 0):
        """mi_unet_debug_capture: run the plan eagerly on `imgs` up to step `layer`; returns (info dict, in, out, pooled, labels)
        of image `img` as float32 NHWC arrays (out = planar logits [classes,H,W] when the step ran the fused head)."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        st = LayerInfo()
        _check(lib().mi_unet_debug_layer_info(self._h, layer, C.byref(st)))
        x = np.empty((st.in_h, st.in_w, st.in_c), np.float32)
        n_out = max(st.out_c, self.cfg.classes)
        y = np.full(st.out_h * st.out_w * n_out, np.nan, np.float32)
        p = np.full((st.out_h // 2, st.out_w // 2, st.out_c), np.nan, np.float32)
        lab = np.full((st.out_h, st.out_w), 255, np.uint8)
        info = LayerInfo()
        _check(lib().mi_unet_debug_capture(self._h, _ptr(imgs), imgs.shape[0], layer, img, _ptr(x), _ptr(y), _ptr(p), _ptr(lab),
                                           C.byref(info)))
        d = info.as_dict()
        if d["skipped"]:
            return d, None, None, None, None
        if d["fused_first"]:             # the step read the u8 image (the first layer ran in its loader): `in` holds the image
            x = x.reshape(-1)[: st.in_h * st.in_w * self.cfg.in_ch].reshape(st.in_h, st.in_w, self.cfg.in_ch).copy()
        if d["fused_head"] or d["kind"] == "head":
            y = y[: self.cfg.classes * st.out_h * st.out_w].reshape(self.cfg.classes, st.out_h, st.out_w)
        else:
            y = y[: st.out_h * st.out_w * st.out_c].reshape(st.out_h, st.out_w, st.out_c)
            lab = None
        return d, x, y, (p if d["pooled"] else None), lab

    def set_profiling(self, on: bool):
        _check(lib().mi_unet_set_profiling(self._h, int(on)))

    def kernel_stats(self, cap=8192):
        n = C.c_int()
        arr = (KernelStat * cap)()
        _check(lib().mi_unet_get_kernel_stats(self._h, arr, cap, C.byref(n)))
        return [dict(name=arr[i].name.decode(), kernel=arr[i].kernel.decode(), flops=arr[i].flops, bytes=arr[i].bytes,
                     ms=arr[i].ms) for i in range(min(n.value, cap))]


def shard_range(n_items: int, rank: int, world: int):
    lo, hi = C.c_int(), C.c_int()
    _check(lib().mi_unet_shard_range(n_items, rank, world, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


class Group:
    """mi_unet_group_*: one engine + worker thread per device in this process, contiguous image shards."""

    GATHER = {"host": 0, "xgmi": 1}

    def __init__(self, height=512, width=512, in_ch=1, base=64, levels=4, classes=3, max_batch=16, devices=None,
                 n_devices=0, conv_algo="auto"):
        self.cfg = Config(height, width, in_ch, base, levels, classes, max_batch, 0, Engine.CONV_ALGOS[conv_algo])
        self._g = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().mi_unet_group_create(C.byref(self.cfg), arr, len(devices), C.byref(self._g)))
        else:
            _check(lib().mi_unet_group_create(C.byref(self.cfg), None, n_devices, C.byref(self._g)))

    def clone(self) -> "Group":
        other = object.__new__(Group)
        other.cfg = self.cfg
        other._g = C.c_void_p()
        _check(lib().mi_unet_group_clone(self._g, C.byref(other._g)))
        return other

    def close(self):
        if self._g:
            lib().mi_unet_group_destroy(self._g)
            self._g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return int(lib().mi_unet_group_size(self._g))

    @property
    def weight_transport(self):
        return lib().mi_unet_group_weight_transport(self._g).decode()

    def load_weights(self, blob: bytes):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _check(lib().mi_unet_group_load_weights_from_memory(self._g, C.cast(buf, C.c_void_p), len(blob)))

    def set_gather(self, mode: str):
        _check(lib().mi_unet_group_set_gather(self._g, self.GATHER[mode]))

    def set_postprocess(self, on: bool):
        _check(lib().mi_unet_group_set_postprocess(self._g, int(on)))

    def infer(self, imgs: np.ndarray, want_logits=False):
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        b, c = imgs.shape[0], self.cfg
        labels = np.empty((b, c.height, c.width), np.uint8)
        logits = np.empty((b, c.classes, c.height, c.width), np.float32) if want_logits else None
        _check(lib().mi_unet_group_infer_u8(self._g, _ptr(imgs), b, _ptr(labels), _ptr(logits)))
        return labels, logits

    def segment_raw16(self, raws, cap_points=8192, cap_contours=64):
        raws = [np.ascontiguousarray(r, dtype=np.uint16) for r in raws]
        n, c = len(raws), self.cfg
        b = n // c.in_ch
        ptrs = (C.c_void_p * n)(*[r.ctypes.data for r in raws])
        ws = (C.c_int * n)(*[r.shape[1] for r in raws])
        hs = (C.c_int * n)(*[r.shape[0] for r in raws])
        tiles = np.empty((b, c.height, c.width) if c.in_ch == 1 else (b, c.height, c.width, c.in_ch), np.uint8)
        masks = np.empty((b, c.height, c.width), np.uint8)
        xy = np.zeros((b, cap_points, 2), np.int32)
        start = np.zeros((b, cap_contours + 1), np.int32)
        counts = np.zeros(b, np.int32)
        _check(lib().mi_unet_group_segment_raw16(self._g, ptrs, ws, hs, b, _ptr(tiles), _ptr(masks), _ptr(xy), cap_points, _ptr(start),
                                                 cap_contours, _ptr(counts)))
        cont = [None if counts[i] < 0 else
                [[tuple(p) for p in xy[i, start[i, k]:start[i, k + 1]].tolist()] for k in range(counts[i])] for i in range(b)]
        return tiles, masks, cont


def layer_debug(op, x, w=None, scale=None, shift=None, relu=False, device=0):
    """Run one layer kernel on host NHWC fp32 data (parity hook)."""
    x = np.ascontiguousarray(x, np.float32)
    b, h, ww, cin = x.shape
    cout = 0
    full_op = op
    if op.endswith("_lpout"):         # 16-bit conv ops: the output tensor is 16-bit on the device too (converted back here)
        op = op[:-6]
    pooled = op.endswith("_pool")     # conv3x3 ops: return the fused 2x2 max-pooled tensor instead of the full-size one
    if pooled:
        op = op[:-5]
    if op in ("conv3x3", "conv3x3_wino", "conv3x3_wino16", "conv3x3_wino4", "conv3x3_wino4s", "conv3x3_wino4a", "conv3x3_wino4b", "conv3x3_bf16", "conv3x3_fp16", "conv3x3_bf16w", "conv3x3_fp16w", "conv3x3_bf16r", "conv3x3_fp16r", "conv3x3_bf16k", "conv3x3_fp16k", "conv3x3_first", "conv3x3_first_bf16", "conv3x3_first_fp16"):
        w = np.ascontiguousarray(w, np.float32)
        cout = w.shape[0]
        out = np.empty((b, h // 2, ww // 2, cout) if pooled else (b, h, ww, cout), np.float32)
    elif op in ("convT2x2", "convT2x2_taps", "convT2x2_bf16", "convT2x2_fp16", "convT2x2_bf16r", "convT2x2_fp16r"):
        w = np.ascontiguousarray(w, np.float32)
        cout = w.shape[1]
        out = np.empty((b, 2 * h, 2 * ww, cout), np.float32)
    elif op == "maxpool":
        out = np.empty((b, h // 2, ww // 2, cin), np.float32)
    else:
        raise ValueError(op)
    scale = None if scale is None else np.ascontiguousarray(scale, np.float32)
    shift = None if shift is None else np.ascontiguousarray(shift, np.float32)
    _check(lib().mi_unet_layer_debug(device, full_op.encode(), _ptr(x), b, h, ww, cin, _ptr(w), _ptr(scale), _ptr(shift), cout,
                                     int(relu), _ptr(out)))
    return out
