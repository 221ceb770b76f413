"""Network spec + weight-file format shared by the oracle, the HIP engine and the tests.

The reference ships no network (its UNet lives in an unpublished TensorRT engine,
/root/reference/src/initialize.cpp:49-60, .gitignore:2-8).  The topology is the one
BASELINE.json names; the free choices are fixed HERE, once:

  * double conv = [conv3x3 pad1 (no bias) -> BatchNorm(eval, eps) -> ReLU] x 2
  * down_i      = maxpool 2x2 s2 -> double conv (C -> 2C)
  * up_i        = convT 2x2 s2 (C -> C/2, with bias) -> concat [skip, upsampled] on channels
                  -> double conv (C -> C/2)
  * outc        = conv1x1 (base -> classes, with bias)
  * I/O contract of the reference: input  fp32 NCHW [B,in_ch,H,W] = u8/255.0f
    (src/process.cpp:22-42, :70-71), output fp32 NCHW planar logits [B,classes,H,W]
    (src/process.cpp:81-85, :163), first-max-wins argmax (src/process.cpp:158-170).

Weight file ("MIUNETW1"), little endian:
    char  magic[8] = "MIUNETW1"
    u32   version  = 1
    u32   in_ch, base, levels, classes
    f32   bn_eps
    u32   n_floats           (payload length, for a truncation check)
    f32   payload[n_floats]  tensors in `tensor_list()` order, PyTorch-native layouts:
          conv3x3 [Cout][Cin][3][3]; BN gamma,beta,mean,var [C]; convT [Cin][Cout][2][2] + bias[Cout];
          outc [classes][base] + bias[classes]
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

MAGIC = b"MIUNETW1"
HEADER = struct.Struct("<8sIIIIIfI")


@dataclass(frozen=True)
class UNetSpec:
    in_ch: int = 1
    base: int = 64
    levels: int = 4
    classes: int = 3
    bn_eps: float = 1e-5

    def channels(self):
        return [self.base << i for i in range(self.levels + 1)]

    def tensor_list(self):
        """[(name, shape)] in file order."""
        out = []

        def dconv(prefix, cin, cout):
            for k, ci in ((1, cin), (2, cout)):
                out.append((f"{prefix}.c{k}.w", (cout, ci, 3, 3)))
                for n in ("gamma", "beta", "mean", "var"):
                    out.append((f"{prefix}.bn{k}.{n}", (cout,)))

        ch = self.channels()
        dconv("inc", self.in_ch, ch[0])
        for i in range(1, self.levels + 1):
            dconv(f"down{i}", ch[i - 1], ch[i])
        for i in range(1, self.levels + 1):
            cin = ch[self.levels - i + 1]
            cout = cin // 2
            out.append((f"up{i}.t.w", (cin, cout, 2, 2)))
            out.append((f"up{i}.t.b", (cout,)))
            dconv(f"up{i}", cin, cout)
        out.append(("outc.w", (self.classes, ch[0])))
        out.append(("outc.b", (self.classes,)))
        return out

    def n_params(self):
        return int(sum(int(np.prod(s)) for _, s in self.tensor_list()))

    def macs_per_image(self, h, w):
        """Algorithmic MACs: conv3x3 = H*W*Cin*Cout*9, convT = Hout*Wout*Cin*Cout, 1x1 = H*W*Cin*Cout."""
        ch = self.channels()
        m = 0
        hh, ww = h, w
        m += hh * ww * 9 * (self.in_ch * ch[0] + ch[0] * ch[0])
        for i in range(1, self.levels + 1):
            hh //= 2
            ww //= 2
            m += hh * ww * 9 * (ch[i - 1] * ch[i] + ch[i] * ch[i])
        for i in range(1, self.levels + 1):
            cin = ch[self.levels - i + 1]
            cout = cin // 2
            hh *= 2
            ww *= 2
            m += hh * ww * cin * cout
            m += hh * ww * 9 * (cin * cout + cout * cout)
        m += hh * ww * ch[0] * self.classes
        return m


def pack_weights(spec: UNetSpec, tensors: dict) -> bytes:
    parts = []
    for name, shape in spec.tensor_list():
        t = np.ascontiguousarray(tensors[name], dtype=np.float32)
        if t.shape != tuple(shape):
            raise ValueError(f"{name}: shape {t.shape} != {shape}")
        parts.append(t.reshape(-1))
    payload = np.concatenate(parts)
    hdr = HEADER.pack(MAGIC, 1, spec.in_ch, spec.base, spec.levels, spec.classes, spec.bn_eps, payload.size)
    return hdr + payload.tobytes()


def unpack_weights(blob: bytes):
    magic, ver, in_ch, base, levels, classes, eps, n = HEADER.unpack_from(blob, 0)
    if magic != MAGIC or ver != 1:
        raise ValueError("not a MIUNETW1 weight file")
    spec = UNetSpec(in_ch, base, levels, classes, eps)
    payload = np.frombuffer(blob, dtype="<f4", count=n, offset=HEADER.size)
    if n != spec.n_params():
        raise ValueError("payload length does not match the header's topology")
    tensors = {}
    off = 0
    for name, shape in spec.tensor_list():
        k = int(np.prod(shape))
        tensors[name] = payload[off:off + k].reshape(shape)
        off += k
    return spec, tensors
