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
