#!/usr/bin/env python3
"""PyTorch state_dict -> MIUNETW1 weight file.

The reference's model chain (.pt -> .onnx -> .trt) is unpublished (/root/reference/.gitignore:2-8); this is the importer a
user of the reference needs to bring their own trained UNet.  Accepted key layout = the common Pytorch-UNet module tree
    inc.double_conv.{0,3}.weight / {1,4}.{weight,bias,running_mean,running_var}
    down{i}.maxpool_conv.1.double_conv....            (i = 1..levels)
    up{i}.up.{weight,bias}, up{i}.conv.double_conv....
    outc.conv.{weight,bias}
(conv biases, if present, are folded into the BatchNorm mean: BN(x + b) == BN'(x) with mean' = mean - b).

    python tools/import_state_dict.py model.pt out.miw [--bn-eps 1e-5]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))
from miunet.spec import UNetSpec, pack_weights  # noqa: E402


def convert(sd: dict, bn_eps: float = 1e-5):
    """sd: name -> array-like.  Returns (spec, weight-file bytes)."""
    g = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32) for k, v in sd.items()
         if not k.endswith("num_batches_tracked")}
    levels = 0
    while f"down{levels + 1}.maxpool_conv.1.double_conv.0.weight" in g:
        levels += 1
    w0 = g["inc.double_conv.0.weight"]
    spec = UNetSpec(in_ch=int(w0.shape[1]), base=int(w0.shape[0]), levels=levels, classes=int(g["outc.conv.weight"].shape[0]),
                    bn_eps=bn_eps)
    t = {}

    def dconv(src, dst):
        for k, (ci, bi) in enumerate(((0, 1), (3, 4)), start=1):
            t[f"{dst}.c{k}.w"] = g[f"{src}.{ci}.weight"]
            mean = g[f"{src}.{bi}.running_mean"].copy()
            if f"{src}.{ci}.bias" in g:
                mean = mean - g[f"{src}.{ci}.bias"]
            t[f"{dst}.bn{k}.gamma"] = g[f"{src}.{bi}.weight"]
            t[f"{dst}.bn{k}.beta"] = g[f"{src}.{bi}.bias"]
            t[f"{dst}.bn{k}.mean"] = mean
            t[f"{dst}.bn{k}.var"] = g[f"{src}.{bi}.running_var"]

    dconv("inc.double_conv", "inc")
    for i in range(1, levels + 1):
        dconv(f"down{i}.maxpool_conv.1.double_conv", f"down{i}")
        t[f"up{i}.t.w"] = g[f"up{i}.up.weight"]
        t[f"up{i}.t.b"] = g[f"up{i}.up.bias"]
        dconv(f"up{i}.conv.double_conv", f"up{i}")
    t["outc.w"] = g["outc.conv.weight"].reshape(spec.classes, spec.base)
    t["outc.b"] = g["outc.conv.bias"]
    return spec, pack_weights(spec, t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("state_dict")
    ap.add_argument("out")
    ap.add_argument("--bn-eps", type=float, default=1e-5)
    a = ap.parse_args()
    import torch

    sd = torch.load(a.state_dict, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    spec, blob = convert(sd, a.bn_eps)
    open(a.out, "wb").write(blob)
    print(f"wrote {a.out}: in_ch={spec.in_ch} base={spec.base} levels={spec.levels} classes={spec.classes} ({len(blob)} bytes)")


if __name__ == "__main__":
    main()
