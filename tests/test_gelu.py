"""Fused tanh-GELU (reference gelu.py:4-12).  CPU: the oracle and the host mirror against the
golden vectors made from the reference module (tests/golden/make_golden_gelu.py), the C ABI's host-side checks.
GPU: the HIP kernel through the C ABI and the torch op against the same goldens and the oracle.

Bar: the kernel evaluates the reference's expression tree one rounded fp32 operation at a time, so the only freedom
is tanhf (device library vs the CPU libm that produced the goldens).  Asserted bound: a 2-ulp difference in tanh
carried through y = (0.5x)(1 + t), i.e. |dy| <= |x| * 1.2e-7 + 2 ulp(y) (1 + t cancels for negative x, so a bound in
ulps of y alone would be wrong); NaN/inf patterns must match exactly."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

import cmhsa
from gelu import GELU, gelu_reference_expression

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gelu_cases  # noqa: E402
from oracle import gelu as oracle  # noqa: E402


def _expected(name):
    return np.load(gelu_cases.golden_path(name))["expected"]


def _assert_close_ulps(got: np.ndarray, want: np.ndarray, x: np.ndarray, ulps: int = 2):
    assert got.shape == want.shape == x.shape
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w)
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf])
    ok = ~(nan_w | inf)
    tol = ulps * np.spacing(np.abs(want[ok]).astype(np.float32)).astype(np.float64) + np.abs(x[ok]).astype(np.float64) * ulps * 6e-8 + 1e-45
    assert np.all(np.abs(got[ok].astype(np.float64) - want[ok].astype(np.float64)) <= tol), \
        float(np.max(np.abs(got[ok].astype(np.float64) - want[ok]) / tol))


@pytest.mark.parametrize("name", list(gelu_cases.GELU_CASES))
def test_oracle_and_host_mirror_equal_the_reference_goldens(name):
    x = torch.from_numpy(gelu_cases.make_input(name))
    want = _expected(name)
    assert np.array_equal(oracle.gelu_tanh_torch(x).numpy(), want, equal_nan=True)
    assert np.array_equal(gelu_reference_expression(x).numpy(), want, equal_nan=True)
    assert np.array_equal(GELU()(x).numpy(), want, equal_nan=True)
    f64 = oracle.gelu_tanh_f64(x.numpy())
    fin = np.isfinite(want) & np.isfinite(f64)
    assert np.max(np.abs(want[fin] - f64[fin]) / np.maximum(1.0, np.abs(f64[fin]))) < 5e-7


def test_reference_edge_values():
    y = oracle.gelu_tanh_torch(torch.tensor([0.0, 1e13, -1e13, float("inf"), float("-inf")]))
    assert y[0] == 0 and y[1] == 1e13 and y[2] == 0 and torch.isinf(y[3]) and torch.isnan(y[4])   # gelu.py:9-12 as written


def test_fused_module_has_no_cpu_fallback():
    with pytest.raises(RuntimeError):
        GELU(fused=True)(torch.zeros(4))


def test_c_abi_argument_checks_need_no_gpu():
    lib = cmhsa.load_library()
    null, good = ctypes.c_void_p(0), ctypes.c_void_p(0x1000)
    assert lib.gpt2_gelu_tanh_f32_fwd(good, good, -1, null) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert lib.gpt2_gelu_tanh_f32_fwd(null, null, 0, null) == cmhsa.CMHSA_OK          # empty input: no-op
    assert lib.gpt2_gelu_tanh_f32_fwd(null, good, 8, null) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert lib.gpt2_gelu_tanh_f32_fwd(good, good, 1 << 40, null) == cmhsa.CMHSA_ERR_TOO_LARGE
    assert lib.gpt2_gelu_algorithmic_bytes(1000) == 8000.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(gelu_cases.GELU_CASES))
def test_kernel_matches_goldens_through_the_c_abi(gpu_device, name):
    x = torch.from_numpy(gelu_cases.make_input(name)).to(gpu_device)
    y = cmhsa.gelu_tanh_fwd(x)
    _assert_close_ulps(y.cpu().numpy(), _expected(name), x.cpu().numpy())
    import causal_multihead_self_attention  # noqa: F401  (in-tree extension module: registers the ops)

    op = torch.ops.causal_multihead_self_attention.gelu_tanh(x)
    assert torch.equal(torch.nan_to_num(op, nan=7.0), torch.nan_to_num(y, nan=7.0))        # same kernel behind the op
    assert torch.equal(torch.nan_to_num(GELU(fused=True)(x), nan=7.0), torch.nan_to_num(y, nan=7.0))


@pytest.mark.gpu
def test_kernel_sizes_alignment_in_place_and_stream(gpu_device):
    torch.manual_seed(3)
    base = torch.randn(1 << 16, device=gpu_device) * 3
    for n in (0, 1, 2, 3, 4, 5, 7, 255, 256, 257, 1023, 4096 + 3, 1 << 16):
        x = base[:n].clone()
        want = oracle.gelu_tanh_torch(x.cpu()).numpy()
        _assert_close_ulps(cmhsa.gelu_tanh_fwd(x).cpu().numpy(), want, x.cpu().numpy())
    # unaligned pointers (4-B aligned only): the scalar path
    x = base[1:1001]
    assert x.data_ptr() % 16 != 0
    out = torch.full((1003,), 9.0, device=gpu_device)[1:1001]
    cmhsa.gelu_tanh_fwd(x.contiguous() if not x.is_contiguous() else x, out=out)
    _assert_close_ulps(out.cpu().numpy(), oracle.gelu_tanh_torch(x.cpu()).numpy(), x.cpu().numpy())
    # in place, and neighbours untouched
    buf = torch.full((4096 + 8,), 5.0, device=gpu_device)
    buf[4:4100] = base[:4096]
    view = buf[4:4100]
    cmhsa.gelu_tanh_fwd(view, out=view)
    _assert_close_ulps(view.cpu().numpy(), oracle.gelu_tanh_torch(base[:4096].cpu()).numpy(), base[:4096].cpu().numpy())
    assert torch.all(buf[:4] == 5.0) and torch.all(buf[4100:] == 5.0)
    # non-default stream
    s = torch.cuda.Stream(gpu_device)
    with torch.cuda.stream(s):
        y = cmhsa.gelu_tanh_fwd(base)
    s.synchronize()
    _assert_close_ulps(y.cpu().numpy(), oracle.gelu_tanh_torch(base.cpu()).numpy(), base.cpu().numpy())


@pytest.mark.gpu
def test_kernel_against_the_unfused_chain_on_the_same_gpu(gpu_device):
    """Inside the model the fused launch replaces PyTorch's eight launches of the same expression on the same GPU.
    Same operation order, every operation rounded separately; PyTorch's device tanh and this library's tanhf are
    different builds of the device math library, so the bound is again 2 ulp of tanh (measured on MI355X: 87 % of the
    elements bit-equal, the rest 1 ulp apart).  Token identity of the generation with the fused kernel is tested in
    tests/test_generation.py."""
    torch.manual_seed(4)
    x = torch.randn(512, 3072, device=gpu_device) * 2
    a = gelu_reference_expression(x)
    b = cmhsa.gelu_tanh_fwd(x)
    _assert_close_ulps(b.cpu().numpy(), a.cpu().numpy(), x.cpu().numpy())


@pytest.mark.gpu
def test_large_input_properties(gpu_device):
    """Size-independent checks at a (B=64, T=512, 4*1600) GPT2-XL activation: gelu(x) - gelu(-x) == x up to rounding,
    saturation y == x for x >= 6, y == 0 (signed) for x <= -6, full coverage of the output buffer."""
    n = 64 * 512 * 6400
    torch.manual_seed(5)
    x = torch.randn(n, device=gpu_device) * 4
    y = torch.full_like(x, float("nan"))
    cmhsa.gelu_tanh_fwd(x, out=y)
    assert not torch.isnan(y).any()
    ym = cmhsa.gelu_tanh_fwd(-x)
    assert ((y - ym) - x).abs().max().item() <= 4e-6
    big = x >= 6
    assert torch.equal(y[big], x[big])
    assert torch.all(y[x <= -6] == 0)
