"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/kprn.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import subprocess

import pytest

from kprn_amd import _ffi, build as kbuild


@pytest.fixture(scope="module")
def so():
    return kbuild.build()


def test_library_builds_and_exports_every_declared_symbol(so):
    assert os.path.exists(so)
    syms = subprocess.check_output(["nm", "-D", so]).decode()
    declared = _ffi.declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if f" T {s}" not in syms]
    assert not missing, missing


def test_library_loads_and_reports_version(so):
    L = ctypes.CDLL(so)
    L.kprn_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.kprn_version()


def test_code_object_is_gfx950(so):
    data = open(so, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_"):
        assert b"amdgcn-amd-amdhsa--" + other not in data


def test_no_gpu_means_a_loud_error_not_a_fallback(so, gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(_ffi.KprnError) as ei:
        _ffi.Engine(6, 100, 9, 4, 8, 4, 16)
    assert ei.value.code in (_ffi.E_DEVICE,)


def test_struct_layouts_match_the_header():
    # field order and sizes of kprn_config / kprn_opt as declared in include/kprn.h
    assert ctypes.sizeof(_ffi.Config) == 20 * 4 + 4 + 4 + 8 + 8  # 20 int32, float, pad to 8, uint64, pointer
    assert ctypes.sizeof(_ffi.Opt) == 12 * 4
    assert ctypes.sizeof(_ffi.ProfEntry) == 48 + 8 + 8


def test_product_code_never_touches_the_oracle():
    root = os.path.dirname(os.path.dirname(__file__))
    for dp, _, files in os.walk(os.path.join(root, "kprn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "from oracle" not in txt and "import oracle" not in txt and "libkprn_oracle" not in txt, f


def test_legacy_default_stream_sentinel_is_declared_everywhere():
    """kprn_config.stream = NULL means "create a stream"; the null stream is asked for by name (round-2 DP ordering bug)"""
    root = os.path.dirname(os.path.dirname(__file__))
    hdr = open(os.path.join(root, "include", "kprn.h")).read()
    assert "#define KPRN_STREAM_LEGACY_DEFAULT ((void*)(intptr_t)-1)" in hdr
    assert _ffi.STREAM_LEGACY_DEFAULT == ctypes.c_void_p(-1).value
    lua = open(os.path.join(root, "bindings", "kprn.lua")).read()
    assert "STREAM_LEGACY_DEFAULT" in lua and "cfg.stream = o.stream" in lua
