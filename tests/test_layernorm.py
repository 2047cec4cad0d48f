"""Fused residual add + LayerNorm (include/gpt2_layernorm.h): oracle pinned against torch's own layer_norm on the CPU;
the HIP kernel against the oracle through the C ABI and through the torch op (GPU)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llm-inference_amd")]

import cmhsa  # noqa: E402
from oracle import layernorm as oracle  # noqa: E402

WIDTHS = (768, 1024, 1280, 1600, 4, 64, 260, 4096)


@pytest.mark.parametrize("d", WIDTHS)
def test_oracle_against_torch_layer_norm_on_cpu(d):
    rng = np.random.default_rng(d)
    x = rng.standard_normal((7, d), dtype=np.float32) * 3 + 1
    delta = rng.standard_normal((7, d), dtype=np.float32)
    w = rng.standard_normal(d, dtype=np.float32)
    b = rng.standard_normal(d, dtype=np.float32)
    s, y = oracle.add_layer_norm_f64(x, delta, w, b)
    ts = torch.from_numpy(x) + torch.from_numpy(delta)
    assert np.array_equal(s, ts.numpy())
    want64 = torch.nn.functional.layer_norm(ts.double(), (d,), torch.from_numpy(w).double(), torch.from_numpy(b).double(), 1e-5)
    assert np.abs(y - want64.numpy()).max() <= 1e-12
    want32 = torch.nn.functional.layer_norm(ts, (d,), torch.from_numpy(w), torch.from_numpy(b), 1e-5)
    assert np.abs(y - want32.numpy()).max() <= 2e-5          # fp32 LayerNorm of values up to ~10: a few ulp of the output scale
    s0, y0 = oracle.add_layer_norm_f64(x, None, w, b)
    assert np.array_equal(s0, x)


def _lib():
    return cmhsa.load_library()


def test_argument_checks_without_a_gpu():
    f = _lib().gpt2_add_layernorm_f32_fwd
    p = ctypes.c_void_p(4096)
    assert f(p, p, p, p, p, p, 0, 768, 768, 768, 1e-5, None) == cmhsa.CMHSA_OK                       # no rows: no-op
    assert f(p, p, p, p, p, p, -1, 768, 768, 768, 1e-5, None) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert f(p, p, p, p, p, p, 4, 770, 770, 770, 1e-5, None) == cmhsa.CMHSA_ERR_BAD_SHAPE            # cols % 4
    assert f(p, p, p, p, p, p, 4, 8192, 8192, 8192, 1e-5, None) == cmhsa.CMHSA_ERR_BAD_SHAPE         # cols > 4096
    assert f(None, p, p, p, p, p, 4, 768, 768, 768, 1e-5, None) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert f(p, p, p, p, None, p, 4, 768, 768, 768, 1e-5, None) == cmhsa.CMHSA_ERR_NULL_POINTER      # delta without sum_out
    assert f(ctypes.c_void_p(4100), p, p, p, p, p, 4, 768, 768, 768, 1e-5, None) == cmhsa.CMHSA_ERR_ALIGNMENT
    assert f(p, p, p, p, p, p, 4, 768, 770, 768, 1e-5, None) == cmhsa.CMHSA_ERR_ALIGNMENT            # stride % 4
    assert f(p, p, p, p, p, p, 1 << 31, 768, 768, 768, 1e-5, None) == cmhsa.CMHSA_ERR_TOO_LARGE


@pytest.mark.gpu
@pytest.mark.parametrize("d", WIDTHS)
@pytest.mark.parametrize("rows", [1, 3, 64, 513])
def test_kernel_against_oracle_through_c_abi(d, rows, gpu_device):
    rng = np.random.default_rng(d * 1000 + rows)
    x = rng.standard_normal((rows, d), dtype=np.float32) * 3 + 1
    delta = rng.standard_normal((rows, d), dtype=np.float32)
    w = rng.standard_normal(d, dtype=np.float32)
    b = rng.standard_normal(d, dtype=np.float32)
    want_s, want_y = oracle.add_layer_norm_f64(x, delta, w, b)
    tx, td, tw, tb = (torch.from_numpy(a).to(gpu_device) for a in (x, delta, w, b))
    s = torch.full((rows, d), float("nan"), device=gpu_device)
    y = torch.full((rows, d), float("nan"), device=gpu_device)
    f = _lib().gpt2_add_layernorm_f32_fwd
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert f(tx.data_ptr(), td.data_ptr(), tw.data_ptr(), tb.data_ptr(), s.data_ptr(), y.data_ptr(), rows, d, d, d, 1e-5, stream) == 0
    assert np.array_equal(s.cpu().numpy(), want_s)                                   # the add is one fp32 operation: exact
    assert np.abs(y.cpu().numpy() - want_y).max() <= 2e-5
    # against torch's own LayerNorm kernel on the same device: the same few ulp
    ty = torch.nn.functional.layer_norm(tx + td, (d,), tw, tb, 1e-5)
    assert (y - ty).abs().max().item() <= 4e-6 * max(1.0, float(ty.abs().max()))
    # without delta: plain LayerNorm of a STRIDED input (every other row), nothing written to sum_out
    y2 = torch.full((rows, d), float("nan"), device=gpu_device)
    wide = torch.from_numpy(np.repeat(x, 2, axis=0)).to(gpu_device)                  # rows at stride 2*d
    assert f(wide.data_ptr(), None, tw.data_ptr(), tb.data_ptr(), None, y2.data_ptr(), rows, d, 2 * d, 0, 1e-5, stream) == 0
    _, want_y2 = oracle.add_layer_norm_f64(x, None, w, b)
    assert np.abs(y2.cpu().numpy() - want_y2).max() <= 2e-5


@pytest.mark.gpu
def test_torch_op_shapes_and_in_model_use(gpu_device):
    import causal_multihead_self_attention  # noqa: F401

    op = torch.ops.causal_multihead_self_attention.add_layer_norm
    torch.manual_seed(3)
    ln = torch.nn.LayerNorm(1600).to(gpu_device)
    with torch.no_grad():
        ln.weight.normal_(); ln.bias.normal_()
    x = torch.randn(4, 37, 1600, device=gpu_device)
    delta = torch.randn(4, 37, 1600, device=gpu_device)
    s, y = op(x, delta, ln.weight, ln.bias, ln.eps)
    assert torch.equal(s, x + delta)
    assert (y - ln(x + delta)).abs().max().item() <= 2e-5
    s0, y0 = op(x, None, ln.weight, ln.bias, ln.eps)
    assert torch.equal(s0, x) and (y0 - ln(x)).abs().max().item() <= 2e-5
    # non-contiguous views (a row picked out of every batch element) are accepted
    xs, ds = x[:, -1:, :], delta[:, -1:, :]
    s1, y1 = op(xs, ds, ln.weight, ln.bias, ln.eps)
    assert s1.shape == (4, 1, 1600) and (y1 - ln(xs + ds)).abs().max().item() <= 2e-5
    with pytest.raises(RuntimeError):
        op(x.cpu(), None, ln.weight, ln.bias, ln.eps)
    with pytest.raises(RuntimeError):
        op(x, delta[:, :5], ln.weight, ln.bias, ln.eps)
