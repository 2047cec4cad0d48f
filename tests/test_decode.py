"""KV-cache decode step (cmhsa_fp32_decode_step, opt-in).  The reference has no KV cache, hence
no fixture of its own: the oracle is the LAST ROW of the reference formula (oracle.decode_step_f64, and the full
attention oracle already pinned by tests/golden), tolerance = north_star's 1e-4 max-abs (observed ~1e-7)."""
import ctypes

import numpy as np
import pytest
import torch

import cmhsa
from oracle import naive_attention as oracle


def test_decode_oracle_is_the_last_row_of_the_full_oracle():
    rng = np.random.default_rng(5)
    for n, H in ((1, 2), (7, 3), (40, 1)):
        q, k, v = (rng.standard_normal((n, 64 * H)).astype(np.float32) for _ in range(3))
        full = oracle.naive_attention_f64(q, k, v, H)
        assert np.abs(oracle.decode_step_f64(q[-1], k, v, H) - full[-1]).max() < 1e-12


def test_decode_argument_validation_needs_no_gpu():
    lib = cmhsa.load_library()
    null, good, bad = ctypes.c_void_p(0), ctypes.c_void_p(0x1000), ctypes.c_void_p(0x1004)

    def call(qkv=good, kc=good, vc=good, o=good, pos=good, qrs=3 * 64, crs=64, cbs=64 * 16, ors=64, B=1, cap=16, H=1, hd=64):
        return lib.cmhsa_fp32_decode_step(qkv, qrs, kc, vc, crs, cbs, o, ors, pos, B, cap, H, hd, null)

    assert call(hd=128) == cmhsa.CMHSA_ERR_HEAD_DIM
    assert call(H=0) == cmhsa.CMHSA_ERR_BAD_SHAPE and call(cap=0) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert call(qrs=2 * 64) == cmhsa.CMHSA_ERR_BAD_SHAPE and call(crs=32) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert call(pos=null) == cmhsa.CMHSA_ERR_NULL_POINTER and call(kc=null) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert call(vc=bad) == cmhsa.CMHSA_ERR_ALIGNMENT and call(crs=66) == cmhsa.CMHSA_ERR_ALIGNMENT
    assert call(B=0) == cmhsa.CMHSA_OK
    assert lib.cmhsa_decode_algorithmic_bytes(2, 100, 12, 64) == 4.0 * 2 * 768 * 206


@pytest.mark.gpu
@pytest.mark.parametrize("B,H", [(1, 12), (3, 25), (2, 1)])
def test_decode_step_equals_last_row_of_full_attention(gpu_device, B, H):
    import causal_multihead_self_attention  # noqa: F401

    op = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_decode
    torch.manual_seed(7)
    d, cap = 64 * H, 1100
    qkv = torch.randn(B, cap, 3 * d, device=gpu_device)
    full = cmhsa.attention_fwd_packed_qkv(qkv, H)                       # (B, cap, d): row t = what decode at pos=t returns
    for pos in (0, 1, 2, 15, 16, 31, 32, 33, 63, 64, 127, 128, 129, 500, 1023, 1099):
        kc = torch.full((B, cap, d), float("nan"), device=gpu_device)   # stale rows are NaN: must never be read
        vc = torch.full((B, cap, d), float("nan"), device=gpu_device)
        kc[:, :pos] = qkv[:, :pos, d:2 * d]
        vc[:, :pos] = qkv[:, :pos, 2 * d:]
        p = torch.tensor([pos], dtype=torch.int32, device=gpu_device)
        new = qkv[:, pos].contiguous()
        out = cmhsa.decode_step(new, kc, vc, p, H)
        assert torch.equal(torch.nan_to_num(op(new, kc.clone(), vc.clone(), p, H)), torch.nan_to_num(out))
        assert (out - full[:, pos]).abs().max().item() <= 2e-6
        # the cache now holds the new token, and nothing else changed
        assert torch.equal(kc[:, pos], qkv[:, pos, d:2 * d]) and torch.equal(vc[:, pos], qkv[:, pos, 2 * d:])
        assert torch.isnan(kc[:, pos + 1:]).all() and torch.isnan(vc[:, pos + 1:]).all()
        assert torch.equal(kc[:, :pos], qkv[:, :pos, d:2 * d])
        if pos in (0, 33, 500):
            want = oracle.decode_step_f64(new[B - 1, :d].cpu().numpy(), kc[B - 1, :pos + 1].cpu().numpy(), vc[B - 1, :pos + 1].cpu().numpy(), H)
            assert np.abs(out[B - 1].cpu().numpy() - want).max() <= 1e-4


@pytest.mark.gpu
def test_decode_out_of_range_pos_zeroes_the_output_only_and_strided_caches_work(gpu_device):
    torch.manual_seed(8)
    H, d, cap = 12, 768, 64
    new = torch.randn(2, 3 * d, device=gpu_device)
    big = torch.zeros(2, cap, 2 * d, device=gpu_device)                 # K and V interleaved in one buffer: row stride 2*d
    kc, vc = big[..., :d], big[..., d:]
    for bad in (-1, cap, cap + 5):                                      # no key exists: o = 0, the caches are not touched
        out = torch.full((2, d), 3.0, device=gpu_device)
        cmhsa.decode_step(new, kc, vc, torch.tensor([bad], dtype=torch.int32, device=gpu_device), H, out=out)
        assert torch.all(out == 0.0) and torch.all(big == 0)
        via_op = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_decode(
            new, kc, vc, torch.tensor([bad], dtype=torch.int32, device=gpu_device), H)
        assert torch.all(via_op == 0.0) and torch.all(big == 0)
    p = torch.zeros(1, dtype=torch.int32, device=gpu_device)
    seq = torch.randn(2, 40, 3 * d, device=gpu_device)
    full = cmhsa.attention_fwd_packed_qkv(seq, H)
    for t in range(40):                                                 # 40 sequential steps build the cache themselves
        p.fill_(t)
        got = cmhsa.decode_step(seq[:, t].contiguous(), kc, vc, p, H)
        assert (got - full[:, t]).abs().max().item() <= 2e-6
    assert torch.equal(kc[:, :40], seq[..., d:2 * d]) and torch.equal(vc[:, :40], seq[..., 2 * d:])
    # non-default stream
    s = torch.cuda.Stream(gpu_device)
    p.fill_(39)
    with torch.cuda.stream(s):
        again = cmhsa.decode_step(seq[:, 39].contiguous(), kc, vc, p, H)
    s.synchronize()
    assert (again - full[:, 39]).abs().max().item() <= 2e-6
