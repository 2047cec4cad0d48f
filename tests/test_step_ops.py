"""The generation loop's bookkeeping kernels (include/gpt2_step.h): embedding sum and greedy next-token step, against
their torch formulations (the reference's own ops: gpt2.py:43-44, sampler.py:14-19)."""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llm-inference_amd")]

import cmhsa  # noqa: E402


def test_argument_checks_without_a_gpu():
    lib = cmhsa.load_library()
    p = ctypes.c_void_p(4096)
    assert lib.gpt2_embed_f32_fwd(p, 8, p, p, p, 0, 8, 768, 50257, None) == cmhsa.CMHSA_OK
    assert lib.gpt2_embed_f32_fwd(p, 8, p, p, p, 2, 8, 770, 50257, None) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert lib.gpt2_embed_f32_fwd(None, 8, p, p, p, 2, 8, 768, 50257, None) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert lib.gpt2_embed_f32_fwd(p, 8, ctypes.c_void_p(4100), p, p, 2, 8, 768, 50257, None) == cmhsa.CMHSA_ERR_ALIGNMENT
    assert lib.gpt2_greedy_step(p, 50257, p, p, 512, p, 0, 50257, 512, 50256, None) == cmhsa.CMHSA_OK
    assert lib.gpt2_greedy_step(p, 50257, p, p, 512, p, 2, 0, 512, 50256, None) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert lib.gpt2_greedy_step(p, 50257, None, p, 512, p, 2, 50257, 512, 50256, None) == cmhsa.CMHSA_ERR_NULL_POINTER
    assert lib.gpt2_greedy_step(p, 50257, p, ctypes.c_void_p(4100), 512, p, 2, 50257, 512, 50256, None) == cmhsa.CMHSA_ERR_ALIGNMENT


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,d", [(1, 8, 768), (3, 64, 768), (8, 512, 1600), (2, 1, 1024), (5, 77, 1280)])
def test_embed_equals_the_torch_expression(B, L, d, gpu_device):
    import causal_multihead_self_attention  # noqa: F401

    torch.manual_seed(L)
    vocab, ctx, cap = 50257, 1024, 576
    wte = torch.randn(vocab, d, device=gpu_device)
    wpe = torch.randn(ctx, d, device=gpu_device)
    buf = torch.randint(0, vocab, (B, cap), device=gpu_device)
    buf[0, 0], buf[-1, L - 1] = 0, vocab - 1
    tokens = buf[:, :L]                                               # a strided view, as in the generation loop
    got = torch.ops.causal_multihead_self_attention.embed_tokens(tokens, wte, wpe)
    want = torch.nn.functional.embedding(tokens, wte) + torch.nn.functional.embedding(torch.arange(L, device=gpu_device), wpe)
    assert got.shape == (B, L, d) and torch.equal(got, want)


@pytest.mark.gpu
def test_greedy_step_equals_the_torch_chain(gpu_device):
    import causal_multihead_self_attention  # noqa: F401

    op = torch.ops.causal_multihead_self_attention.greedy_step
    torch.manual_seed(9)
    B, vocab, cap, eos = 6, 50257, 96, 50256
    for trial in range(6):
        logits = torch.randn(B, vocab, device=gpu_device)
        logits[0, 777] = logits[0, 40000] = 50.0                      # a tie: the FIRST maximum wins
        logits[1, eos] = 60.0                                         # this row generates EOS now
        logits[2, 0] = 70.0                                           # first element
        logits[3, vocab - 1] = 70.0                                   # last element (that is EOS as well)
        logits[4, 123] = float("nan"); logits[4, 5] = 45.0            # NaN ranks above every number (torch.argmax)
        logits[4, 30000] = float("nan")                               # ... and the FIRST NaN wins
        done = torch.zeros(B, dtype=torch.bool, device=gpu_device)
        done[5] = True                                                # a finished row keeps receiving EOS
        tokens = torch.randint(0, vocab, (B, cap), device=gpu_device)
        pos = torch.tensor([10 + trial], device=gpu_device)
        # the torch chain of BucketedGreedyGenerator._step = reference sampler.py:17
        nxt = torch.argmax(logits, dim=-1)
        nxt = torch.where(done, torch.full_like(nxt, eos), nxt)
        want_done = done | (nxt == eos)
        want_tokens = tokens.clone()
        want_tokens[:, int(pos) + 1] = nxt
        view = logits.unsqueeze(1)[:, 0, :]                           # (B, vocab) view with a row stride, as in the loop
        op(view, done, tokens, pos, eos)
        assert torch.equal(tokens, want_tokens) and torch.equal(done, want_done)
        assert int(pos) == 10 + trial                                 # pos is the caller's to increment
        assert tokens[0, int(pos) + 1].item() == 777 and tokens[4, int(pos) + 1].item() == 123
    # a row of NaNs picks index 0, a row of -inf as well (torch.argmax's answers)
    weird = torch.full((2, vocab), float("nan"), device=gpu_device)
    weird[1] = float("-inf")
    d2 = torch.zeros(2, dtype=torch.bool, device=gpu_device)
    t2 = torch.full((2, cap), 7, dtype=torch.long, device=gpu_device)
    op(weird, d2, t2, torch.tensor([3], device=gpu_device), eos)
    assert t2[:, 4].tolist() == [0, 0] and not d2.any()
    # writing past the buffer is refused silently (nothing written)
    before = tokens.clone()
    op(logits, done, tokens, torch.tensor([cap - 1], device=gpu_device), eos)
    assert torch.equal(tokens, before)
