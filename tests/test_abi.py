"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/*.h declare, and its host-only entry points behave.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

import cmhsa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))


def _declared_functions():
    names = set()
    for header in HEADERS:
        text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
        names |= set(re.findall(r"\b((?:cmhsa|gpt2)_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_and_python_symbol_lists_agree():
    assert _declared_functions() == sorted(cmhsa.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = cmhsa.load_library()
    for name in _declared_functions():
        assert getattr(lib, name) is not None, name


def test_no_torch_symbols_in_kernel_library():
    # the boundary is plain C: the kernel library must not depend on libtorch / libc10
    import subprocess

    out = subprocess.run(["readelf", "-d", cmhsa.LIB_PATH], capture_output=True, text=True, check=True).stdout
    needed = re.findall(r"Shared library: \[(.*?)\]", out)
    assert not [n for n in needed if "torch" in n or "c10" in n], needed


def test_version_and_status_strings():
    lib = cmhsa.load_library()
    assert lib.cmhsa_version() >> 16 == 1
    assert cmhsa.status_string(cmhsa.CMHSA_OK) == "success"
    assert "64" in cmhsa.status_string(cmhsa.CMHSA_ERR_HEAD_DIM)
    for code in (-1, -2, -3, -4, -5):
        assert cmhsa.status_string(code)


def test_algorithmic_flops_matches_survey_figures():
    # config 2 = 1.612 GFLOP, config 5 = 1099.8 GFLOP
    assert cmhsa.algorithmic_flops(1, 1024, 12, 64) == 2 * 12 * 64 * 1024 * 1025
    assert abs(cmhsa.algorithmic_flops(32, 4096, 16, 64) / 1e9 - 1099.8) < 0.1


def test_argument_validation_needs_no_gpu():
    # every rejection below happens before any HIP call
    lib = cmhsa.load_library()
    null = ctypes.c_void_p(0)
    good = ctypes.c_void_p(0x1000)
    bad_align = ctypes.c_void_p(0x1004)

    def call(q=good, k=good, v=good, o=good, rs=(64, 64, 64, 64), bs=(0, 0, 0, 0), B=1, T=8, H=1, hd=64):
        return lib.cmhsa_fp32_fwd(q, k, v, o, *rs, *bs, B, T, H, hd, null)

    assert call(hd=32) == cmhsa.CMHSA_ERR_HEAD_DIM
    assert call(H=0) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert call(T=-1) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert call(q=null) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert call(k=bad_align) == cmhsa.CMHSA_ERR_ALIGNMENT
    assert call(rs=(64, 66, 64, 64)) == cmhsa.CMHSA_ERR_ALIGNMENT
    assert call(rs=(64, -64, 64, 64)) == cmhsa.CMHSA_ERR_BAD_SHAPE          # negative strides are not supported
    assert call(T=0) == cmhsa.CMHSA_OK          # empty sequence: successful no-op
    assert call(B=0) == cmhsa.CMHSA_OK
    # the kernel forms 32-bit row numbers up to (ceil(T/32) + 16) * 32: a seq_len just over that bound is rejected
    # explicitly, one far below it passes this check (and is then rejected for its grid size -- also before any launch)
    assert call(T=2**31 - 1) == cmhsa.CMHSA_ERR_TOO_LARGE
    assert call(T=2**31 - 512) == cmhsa.CMHSA_ERR_TOO_LARGE
    assert call(rs=(2**24 + 4, 64, 64, 64)) == cmhsa.CMHSA_ERR_TOO_LARGE      # 32 rows * stride * 4 B must fit 32-bit offsets
    assert lib.cmhsa_fp32_fwd_2d(good, good, good, good, 100, 100, 100, 100, 8, 100, 3, null) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert lib.cmhsa_fp32_fwd_2d(good, good, good, good, 96, 96, 96, 96, 8, 96, 3, null) == cmhsa.CMHSA_ERR_HEAD_DIM


def test_extension_module_is_in_tree():
    import sysconfig

    ext = os.path.join(ROOT, "llm-inference_amd", "causal_multihead_self_attention" + sysconfig.get_config_var("EXT_SUFFIX"))
    assert os.path.exists(ext), "run python llm-inference_amd/build.py"


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(cmhsa, "_lib", None)
    monkeypatch.setattr(cmhsa, "LIB_PATH", "/nonexistent/libcmhsa_hip.so")
    with pytest.raises(FileNotFoundError):
        cmhsa.load_library()


def test_packed_qkv_argument_validation_needs_no_gpu():
    lib = cmhsa.load_library()
    null, good = ctypes.c_void_p(0), ctypes.c_void_p(0x1000)
    call = lambda qkv=good, o=good, rs=3 * 64, ors=64, T=8, H=1, hd=64: lib.cmhsa_fp32_fwd_packed_qkv(qkv, o, rs, ors, 0, 0, 1, T, H, hd, null, 0, null)
    assert call(hd=32) == cmhsa.CMHSA_ERR_HEAD_DIM
    assert call(H=0) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert call(rs=2 * 64) == cmhsa.CMHSA_ERR_BAD_SHAPE          # a row must hold Q | K | V
    assert call(qkv=null) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert call(rs=3 * 64 + 2) == cmhsa.CMHSA_ERR_ALIGNMENT
    assert call(T=0) == cmhsa.CMHSA_OK
