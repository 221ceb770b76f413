"""f4 helpers on the CPU: the PyTorch state_dict importer and the REPL's command handling (no engine needed)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as orc
from miunet import synth
from miunet.spec import UNetSpec, pack_weights, unpack_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CLI = os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd", "medseg_cli")


def _torch_style_state_dict(spec, t, conv_bias=False, rng=None):
    sd = {}

    def dconv(src, dst):
        for k, (ci, bi) in enumerate(((0, 1), (3, 4)), start=1):
            sd[f"{dst}.{ci}.weight"] = t[f"{src}.c{k}.w"]
            mean = t[f"{src}.bn{k}.mean"].copy()
            if conv_bias:                                    # BN(conv(x) + b): mean' = mean - b  <=>  mean = mean' + b
                b = rng.standard_normal(mean.shape).astype(np.float32) * 0.1
                sd[f"{dst}.{ci}.bias"] = b
                mean = mean + b
            sd[f"{dst}.{bi}.weight"] = t[f"{src}.bn{k}.gamma"]
            sd[f"{dst}.{bi}.bias"] = t[f"{src}.bn{k}.beta"]
            sd[f"{dst}.{bi}.running_mean"] = mean
            sd[f"{dst}.{bi}.running_var"] = t[f"{src}.bn{k}.var"]
            sd[f"{dst}.{bi}.num_batches_tracked"] = np.array(7)

    dconv("inc", "inc.double_conv")
    for i in range(1, spec.levels + 1):
        dconv(f"down{i}", f"down{i}.maxpool_conv.1.double_conv")
        sd[f"up{i}.up.weight"] = t[f"up{i}.t.w"]
        sd[f"up{i}.up.bias"] = t[f"up{i}.t.b"]
        dconv(f"up{i}", f"up{i}.conv.double_conv")
    sd["outc.conv.weight"] = t["outc.w"].reshape(spec.classes, spec.base, 1, 1)
    sd["outc.conv.bias"] = t["outc.b"]
    return sd


def test_state_dict_importer_roundtrip():
    import import_state_dict as imp

    spec = UNetSpec(1, 16, 3, 3)
    t = synth.make_weights(spec, 9)
    spec2, blob = imp.convert(_torch_style_state_dict(spec, t))
    assert (spec2.in_ch, spec2.base, spec2.levels, spec2.classes) == (1, 16, 3, 3)
    assert blob == pack_weights(spec, t)
    # conv biases are folded into the BatchNorm mean: the network function is unchanged
    rng = np.random.default_rng(1)
    _, blob_b = imp.convert(_torch_style_state_dict(spec, t, conv_bias=True, rng=rng))
    _, tb = unpack_weights(blob_b)
    for k in t:
        assert np.allclose(tb[k], t[k], atol=1e-6), k
    imgs = synth.make_images(1, 16, 24, 1, 3)
    a, _ = orc.unet_forward(blob, imgs)
    b, _ = orc.unet_forward(blob_b, imgs)
    assert np.max(np.abs(a - b)) < 1e-5


@pytest.mark.skipif(not os.path.exists(CLI), reason="medseg_cli not built")
def test_cli_command_handling_without_an_engine(tmp_path):
    script = "help\nprocess /nowhere 4 4\ninit\ninit {}\nfrobnicate\nexit\n".format(tmp_path / "engine" / "missing.miw")
    r = subprocess.run([CLI], input=script.encode(), capture_output=True, timeout=60)
    out, err = r.stdout.decode(), r.stderr.decode()
    assert r.returncode == 0
    assert "Welcome to Medical Image Segmentation Tool" in out and out.count("Commands:") == 2 and "Exiting..." in out
    assert "Error: Engine not initialized" in err                      # process before init (src/main.cpp:97-100)
    assert "Error: Missing weight file path" in err
    assert "Engine initialization failed" in err                       # missing file -> initialize_engine returns false
    assert "Unknown command: frobnicate" in err
    # log dir derived as <dir of engine>/../log (src/main.cpp:87) and the banner + error lines are in the log
    log = (tmp_path / "engine" / ".." / "log" / "segmentation_log.txt").read_text()
    assert "=== Initializing Medical Image Segmentation Engine ===" in log and "not found" in log


def test_lds_layouts_of_the_16bit_kernels_are_conflict_free():
    """tools/dev/lds_bank_model.py: gfx950 services a ds_read_b128 in four groups of 16 lanes over 64 banks; the piece permutation of
    csrc/lpr_common.h (lds_swz_row16) must put every fragment read of the 16-bit kernels at 4 LDS cycles, for every patch row,
    column half and tap displacement -- and the model must still see the 2-way conflict of the round-2 layout it replaced (the
    counters of profiles/r03_ab_lds_swizzle.txt agree with both)."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
    import lds_bank_model as m
    for name, fn in list(m.SHIPPED_16BIT.items()) + list(m.SHIPPED_FP32.items()):
        mean, worst = fn()
        assert mean == 4.0 and worst == 4, (name, mean, worst)
    assert m.round2_frag32_rows2() == (8.0, 8) and m.round2_frag32_row1() == (8.0, 8) and m.round2_wino4_v() == (8.0, 8)
    # the kernels' permutation is the model's: one source line each
    src = open(os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd", "csrc", "lpr_common.h")).read()
    assert "lds_swz_row16(int col) { return 2 * ((col >> 2) & 1); }" in src
    assert all(m.swz_row16(c) == 2 * ((c >> 2) & 1) for c in range(64))
