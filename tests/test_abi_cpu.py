"""CPU-side checks of the C-ABI: the library loads, exports every symbol include/mi_unet.h declares, and fails
LOUDLY (no CPU fallback) when no HIP device is present.  No compute calls."""
import ctypes
import os
import re

import pytest

from miunet import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mi_unet.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_unet_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = binding.lib()
    syms = header_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(L, s), f"libmiunet.so does not export {s}"
    assert sorted(binding.EXPORTS) == syms


def test_default_config_is_the_reference_constants():
    c = binding.default_config()
    # src/process.cpp:70 (1x1x512x512), :162 (3 classes)
    assert (c.height, c.width, c.in_ch, c.classes) == (512, 512, 1, 3)
    assert (c.base, c.levels) == (64, 4)


def test_no_device_means_loud_failure_not_fallback():
    if binding.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(binding.MiUnetError) as ei:
        binding.Engine(height=64, width=64, max_batch=1)
    assert ei.value.code == 2            # MI_UNET_ENODEVICE
    assert "no CPU fallback" in str(ei.value)


def test_null_handle_is_an_error():
    L = binding.lib()
    assert L.mi_unet_sync(None) != 0
    assert b"null engine handle" in L.mi_unet_last_error()


def test_staging_copy_pool_copies_every_byte(tmp_path):
    """csrc/copy_pool.h (the pageable -> pinned staging copy of the RAW-in entry points and the tiles / masks copy-out) against
    memcpy for 2..16 parts over sizes in [1 MB, 1 MB + 64 KB]: the floored piece size of round 3 dropped the last bytes of any
    image whose size / parts was a whole number of pages -- a stale sample in the pinned ring then changed the
    min/max normalisation of the whole tile."""
    import subprocess

    exe = tmp_path / "copy_pool_test"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(ROOT, "tests", "cpu", "copy_pool_test.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    assert b"all copy pool checks passed" in r.stdout


def test_product_library_has_no_experiment_kernels():
    """The reference's engine has exactly one behaviour (src/process.cpp:147).  The timing-only kernel instantiations (results
    WRONG by design) and the environment switches that select them exist only in the lab build (`make lab`: libmiunet_exp.so,
    -DMIUNET_EXPERIMENTS); libmiunet.so must contain neither the instantiations nor the variable names."""
    import subprocess

    so = binding.LIB_PATH
    syms = subprocess.run(["nm", "-C", so], capture_output=True, check=True).stdout.decode()
    bad = []
    for line in syms.splitlines():
        m = re.search(r"conv3x3_wino4_f32<\d+, \w+, \w+, (\d+)>", line)
        if m and m.group(1) != "0":
            bad.append(line)
        m = re.search(r"conv3x3_wino4s_f32<\w+, (\d+), (\d+), (\w+)>", line)
        if m and (m.group(2) != "0" or m.group(1) != "6" or m.group(3) != "true"):
            bad.append(line)
        m = re.search(r"conv3x3_lp2<.*, (\d+), (\d+)>", line)
        if m and (m.group(1) != "0" or m.group(2) != "6"):
            bad.append(line)
    assert not bad, "\n".join(bad[:10])
    assert "conv3x3_wino4_f32<2, false, false, 0>" in syms          # the check above looked at real names
    blob = open(so, "rb").read()
    for name in (b"MIUNET_W4_EXP", b"MIUNET_W4S_EXP", b"MIUNET_LP2_EXP", b"MIUNET_LPR_EXP", b"MIUNET_WINO4S_ONE_WG", b"MIUNET_W4S_UD",
                 b"MIUNET_LP2_WD", b"MIUNET_LP2_NSPLIT", b"MIUNET_LP2_MINCIN", b"MIUNET_CONVT_CFG", b"MIUNET_CONVT_WPS", b"MIUNET_FIRST_RBW", b"MIUNET_WINO4A_HSACO", b"MIUNET_WINO4B_HSACO"):
        assert name not in blob, name.decode()
