"""The no-allocation entry point cmhsa_fp32_fwd_noalloc (include/cmhsa.h): cmhsa_fp32_fwd_ws with the workspace in a
static device array of the library.  Called through the C ABI and compared, bit for bit, with cmhsa_fp32_fwd_ws (the
existing entry point whose kernels use the scratch buffer) -- same plan, same arithmetic -- on two calls in a row, the
second with a SMALLER input (the shared array keeps the first call's partials; nothing of them may leak), then once
more with the first input; plus the f64 oracle, the size limit and the argument checks.

Collected last on purpose (file name): the round-end run is `pytest -x`."""
import ctypes

import numpy as np
import pytest
import torch

import cmhsa
from oracle import naive_attention as oracle

pytestmark = pytest.mark.gpu


def _args(q, k, v, o, B, T, H):
    return (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), q.stride(1), k.stride(1), v.stride(1), o.stride(1),
            q.stride(0), k.stride(0), v.stride(0), o.stride(0), B, T, H, 64)


def _inputs(B, T, H, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = 64 * H
    x = torch.randn((B, T, 3 * d), generator=g).to(device)
    return x[..., :d], x[..., d:2 * d], x[..., 2 * d:]


def test_noalloc_entry_point_equals_workspace_entry_point(gpu_device):
    lib = cmhsa.load_library()
    stream = ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    ws = cmhsa.Workspace(gpu_device)
    big = (1, 1024, 12)          # BASELINE config 2: four query blocks per head split across workgroups
    small = (1, 768, 12)         # a smaller input through the same array: split as well, fewer and other parts (T = 512 and
                                 # below run on the 16-key-tile family, which needs no scratch)
    for shape in (big, small):
        need = lib.cmhsa_fp32_fwd_workspace_bytes(*shape)
        assert 0 < need <= cmhsa.NOALLOC_SCRATCH_BYTES, (shape, need)      # both really use the scratch buffer
    for i, (B, T, H) in enumerate((big, small, big)):
        q, k, v = _inputs(B, T, H, 50 + i, gpu_device)
        want = torch.full((B, T, 64 * H), float("nan"), device=gpu_device)
        got = torch.full((B, T, 64 * H), float("nan"), device=gpu_device)
        assert lib.cmhsa_fp32_fwd_ws(*_args(q, k, v, want, B, T, H), ws.ptr, ws.nbytes, stream) == 0
        assert lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, got, B, T, H), stream) == 0
        torch.cuda.synchronize(gpu_device)
        assert torch.isfinite(got).all()
        assert torch.equal(got, want), (B, T, H)
        for h in (0, H - 1):
            sl = slice(64 * h, 64 * h + 64)
            ref = oracle.naive_attention_f64(q[0, :, sl].cpu().numpy(), k[0, :, sl].cpu().numpy(), v[0, :, sl].cpu().numpy(), 1)
            assert np.abs(got[0, :, sl].cpu().numpy() - ref).max() <= 1e-4
    # back to back, no host synchronisation in between, alternating sizes: the array's counters are left at zero by
    # the kernel itself
    qa, ka, va = _inputs(*big, 60, gpu_device)
    qb, kb, vb = _inputs(*small, 61, gpu_device)
    wa = cmhsa.attention_fwd(qa, ka, va, 12, workspace=ws)
    wb = cmhsa.attention_fwd(qb, kb, vb, 12, workspace=ws)
    outs = []
    for i in range(20):
        q, k, v, shape = (qa, ka, va, big) if i % 2 == 0 else (qb, kb, vb, small)
        o = torch.empty((shape[0], shape[1], 768), device=gpu_device)
        assert lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, o, *shape), stream) == 0
        outs.append(o)
    torch.cuda.synchronize(gpu_device)
    for i, o in enumerate(outs):
        assert torch.equal(o, wa if i % 2 == 0 else wb), i


def test_noalloc_entry_point_limits_and_checks(gpu_device):
    lib = cmhsa.load_library()
    stream = ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    # a shape whose plan needs no scratch at all runs like cmhsa_fp32_fwd
    q, k, v = _inputs(2, 100, 3, 70, gpu_device)
    o = torch.empty((2, 100, 192), device=gpu_device)
    assert lib.cmhsa_fp32_fwd_workspace_bytes(2, 100, 3) == 0
    assert lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, o, 2, 100, 3), stream) == 0
    assert torch.equal(o, cmhsa.attention_fwd(q, k, v, 3))
    # a shape whose plan needs more than the static array is rejected, nothing is launched
    too_big = next((s for s in ((1, 1536, 5), (1, 2048, 4), (1, 1024, 7), (2, 2048, 2), (1, 1536, 4))
                    if lib.cmhsa_fp32_fwd_workspace_bytes(*s) > cmhsa.NOALLOC_SCRATCH_BYTES), None)
    assert too_big is not None, "no probe shape exceeds the static scratch any more: pick new ones"
    B, T, H = too_big
    q, k, v = _inputs(B, T, H, 71, gpu_device)
    o = torch.full((B, T, 64 * H), 7.0, device=gpu_device)
    assert lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, o, B, T, H), stream) == cmhsa.CMHSA_ERR_WORKSPACE
    torch.cuda.synchronize(gpu_device)
    assert bool((o == 7.0).all())
    assert "CMHSA_NOALLOC_SCRATCH_BYTES" in cmhsa.status_string(cmhsa.CMHSA_ERR_WORKSPACE)
    # the usual argument checks
    assert lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, o, B, T, H)[:15], 32, stream) == cmhsa.CMHSA_ERR_HEAD_DIM
    assert lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, o, B, 0, H), stream) == 0


def test_noalloc_entry_point_captured_in_a_hip_graph(gpu_device):
    """What the entry point is for: a call under HIP graph capture.  One un-captured call per (device, shape) first (it takes
    the scratch array's address, raises the kernel's LDS limit and builds the shape's plan: host-side set-up that does not
    belong in a capture); then the same call captured on a side stream, replayed three times on changing inputs: every replay
    bit-equal to the eager workspace call on the same inputs."""
    lib = cmhsa.load_library()
    B, T, H = 1, 1024, 12
    ws = cmhsa.Workspace(gpu_device)
    q, k, v = (t.contiguous() for t in _inputs(B, T, H, 80, gpu_device))
    out = torch.empty((B, T, 64 * H), device=gpu_device)
    side = torch.cuda.Stream(gpu_device)
    with torch.cuda.stream(side):
        assert lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, out, B, T, H), ctypes.c_void_p(side.cuda_stream)) == 0     # the warm-up call
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = lib.cmhsa_fp32_fwd_noalloc(*_args(q, k, v, out, B, T, H), ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream))
    assert rc == 0
    for seed in (81, 82, 83):
        nq, nk, nv = _inputs(B, T, H, seed, gpu_device)
        q.copy_(nq); k.copy_(nk); v.copy_(nv)
        out.fill_(float("nan"))
        torch.cuda.synchronize(gpu_device)
        graph.replay()
        torch.cuda.synchronize(gpu_device)
        want = torch.empty_like(out)
        stream = ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
        assert lib.cmhsa_fp32_fwd_ws(*_args(q, k, v, want, B, T, H), ws.ptr, ws.nbytes, stream) == 0
        torch.cuda.synchronize(gpu_device)
        assert torch.equal(out, want), seed

