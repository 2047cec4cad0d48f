"""Few-row linear layer (include/gpt2_linear.h): oracle pinned against torch.nn.functional.linear on the CPU; the HIP kernel
against the oracle through the C ABI and through the torch op (GPU)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llm-inference_amd")]

import cmhsa  # noqa: E402
from oracle import gelu as gelu_oracle  # noqa: E402
from oracle import linear as oracle  # noqa: E402

# (n_out, k): GPT2-small's and GPT2-XL's four layers, the LM head (narrowed), and ragged sizes
LAYERS = ((2304, 768), (768, 768), (3072, 768), (768, 3072), (4800, 1600), (1600, 6400), (5003, 768), (1, 4), (7, 20), (130, 4100), (33, 16388))


@pytest.mark.parametrize("n_out,k", LAYERS[:4] + LAYERS[6:9])
def test_oracle_against_torch_linear_on_cpu(n_out, k):
    rng = np.random.default_rng(n_out + k)
    x = rng.standard_normal((3, k), dtype=np.float32)
    w = rng.standard_normal((n_out, k), dtype=np.float32) * 0.05
    b = rng.standard_normal(n_out, dtype=np.float32)
    y = oracle.linear_f64(x, w, b)
    want64 = torch.nn.functional.linear(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double())
    assert np.abs(y - want64.numpy()).max() <= 1e-12
    want32 = torch.nn.functional.linear(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b))
    assert (np.abs(y - want32.numpy()) <= 1e-6 * oracle.sum_scale(x, w, b) + 1e-7).all()
    # with the activation: the reference's GELU module formula on the fp32 linear output
    g = oracle.linear_f64(x, w, b, gelu=True)
    want_g = gelu_oracle.gelu_tanh_torch(want32).numpy()
    assert np.abs(g - want_g).max() <= 1e-5
    assert np.abs(oracle.linear_f64(x, w) - (want64 - torch.from_numpy(b).double()).numpy()).max() <= 1e-12


def test_argument_checks_without_a_gpu():
    f = cmhsa.load_library().gpt2_linear_rows_f32_fwd
    p = ctypes.c_void_p(4096)
    assert f(p, 768, p, 768, p, p, 2304, 0, 2304, 768, 0, None) == cmhsa.CMHSA_OK                    # no rows: no-op
    assert f(p, 768, p, 768, p, p, 2304, 1, 0, 768, 0, None) == cmhsa.CMHSA_OK                       # no outputs: no-op
    assert f(p, 768, p, 768, p, p, 2304, 9, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_BAD_SHAPE         # more than 8 rows
    assert f(p, 768, p, 768, p, p, 2304, -1, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert f(p, 770, p, 770, p, p, 2304, 1, 2304, 770, 0, None) == cmhsa.CMHSA_ERR_BAD_SHAPE         # k % 4
    assert f(p, 768, p, 768, p, p, 2304, 1, 2304, 768, 7, None) == cmhsa.CMHSA_ERR_BAD_SHAPE         # unknown activation
    assert f(p, 764, p, 768, p, p, 2304, 2, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_BAD_SHAPE         # rows of x overlap
    assert f(p, 768, p, 764, p, p, 2304, 2, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert f(p, 768, p, 768, p, p, 2300, 2, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert f(None, 768, p, 768, p, p, 2304, 1, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert f(p, 768, p, 768, p, None, 2304, 1, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert f(ctypes.c_void_p(4100), 768, p, 768, p, p, 2304, 1, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_ALIGNMENT
    assert f(p, 770, p, 768, p, p, 2304, 1, 2304, 768, 0, None) == cmhsa.CMHSA_ERR_ALIGNMENT         # stride % 4
    assert cmhsa.load_library().gpt2_linear_rows_algorithmic_bytes(1, 2304, 768, 1) == 4.0 * (2304 * 768 + 2304 + 768 + 2304)


@pytest.mark.gpu
@pytest.mark.parametrize("n_out,k", LAYERS)
@pytest.mark.parametrize("rows", [1, 2, 3, 5, 8])
def test_kernel_against_oracle_through_c_abi(n_out, k, rows, gpu_device):
    rng = np.random.default_rng(n_out * 7 + k + rows)
    x = rng.standard_normal((rows, k), dtype=np.float32)
    w = (rng.standard_normal((n_out, k), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(n_out, dtype=np.float32)
    scale = oracle.sum_scale(x, w, b)
    tx, tw, tb = (torch.from_numpy(a).to(gpu_device) for a in (x, w, b))
    stream = ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    f = cmhsa.load_library().gpt2_linear_rows_f32_fwd
    for act in (0, 1):
        for bias in (tb, None):
            y = torch.full((rows, n_out + 3), float("nan"), device=gpu_device)         # padded rows: a row stride of its own
            rc = f(tx.data_ptr(), k, tw.data_ptr(), k, bias.data_ptr() if bias is not None else None, y.data_ptr(), n_out + 3,
                   rows, n_out, k, act, stream)
            assert rc == 0
            got = y.cpu().numpy()
            assert np.isnan(got[:, n_out:]).all()                                      # nothing written past a row's end
            lin = oracle.linear_f64(x, w, b if bias is not None else None)
            want = gelu_oracle.gelu_tanh_f64(lin) if act else lin
            # fp32 sum of k terms in a tree of 64 strided partials: a few roundings of the sum of magnitudes (GELU's slope is <= ~1.13)
            assert (np.abs(got[:, :n_out] - want) <= 2e-6 * scale + 1e-6).all(), (act, bias is not None)


@pytest.mark.gpu
def test_strided_weight_rows_and_torch_op(gpu_device):
    """weight as a row-strided view (rows of a wider matrix), x with leading dimensions, through the torch op; equal to the
    C-ABI result bit for bit and within fp32 rounding of torch's own linear; more than 8 rows are refused."""
    import causal_multihead_self_attention  # noqa: F401

    op = torch.ops.causal_multihead_self_attention.linear_rows
    torch.manual_seed(3)
    wide = torch.randn(1000, 1024, device=gpu_device) * 0.05
    w = wide[:, :768]                                                  # row stride 1024
    b = torch.randn(1000, device=gpu_device)
    x = torch.randn(2, 1, 768, device=gpu_device)
    y = op(x, w, b, 0)
    assert y.shape == (2, 1, 1000)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert (y.double() - ref).abs().max().item() <= 2e-5
    yg = op(x, w, b, 1)
    assert torch.equal(yg, torch.ops.causal_multihead_self_attention.gelu_tanh(y))     # the same GELU code on the same sums
    y1 = op(x[0, 0], w, None, 0)
    assert y1.shape == (1000,) and (y1.double() - (ref[0, 0] - b.double())).abs().max().item() <= 2e-5
    with pytest.raises(RuntimeError, match="at most 8 rows"):
        op(torch.randn(9, 768, device=gpu_device), w, b, 0)
    # the same row of x gives the same bits whatever the number of rows beside it
    xs = torch.randn(8, 768, device=gpu_device)
    all8 = op(xs, w, b, 0)
    for r in (1, 2, 4, 5):
        assert torch.equal(op(xs[:r], w, b, 0), all8[:r])
