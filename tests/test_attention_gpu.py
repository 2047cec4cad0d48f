"""GPU parity tests (run with -m gpu on an MI355X): the hand-written HIP kernel, called through the
C ABI (ctypes -> cmhsa_fp32_fwd) and through the torch op, against
  * the golden vectors made by the reference module (tests/golden/*.npz),
  * the CPU oracle (oracle/naive_attention.py) on the same seeded inputs,
  * size-independent properties at the BASELINE.json sizes.

Tolerance: north_star states max-abs-error <= 1e-4 in fp32.  The
reference's own test uses 1e-6 on U[0,1) inputs (test_torch_extension.py:7); we assert the
1e-4 contract everywhere and additionally hold the U[0,1) sweep to 2e-6 (the reference formula
itself sits 1e-6 away from an fp64 evaluation at T=1024, see tests/golden/make_golden.py output).
"""
import numpy as np
import pytest
import torch

import cases
import cmhsa
from causal_self_attention import (CausalSelfAttention, ScaledDotProductAttentionBackend,
                                   scaled_dot_product_attention)
from oracle import naive_attention as oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4            # north_star contract
TOL_RAND = 2e-6       # U[0,1) goldens up to T=1024 (the reference formula itself sits ~1e-6 from an f64 evaluation there)
TOL_REF = 1e-6        # the reference's own bound on its sweep (test_torch_extension.py:7, :22: seq_len 1..199, U[0,1))
CUSTOM = ScaledDotProductAttentionBackend.CUSTOM_CUDA
NAIVE = ScaledDotProductAttentionBackend.NAIVE_PYTORCH


def _dev(arrs, device):
    """numpy (possibly strided views of one buffer) -> device tensors with the SAME strides."""
    base = arrs[0].base
    if base is not None and all(a.base is base for a in arrs):
        buf = torch.from_numpy(base).to(device)
        d = arrs[0].shape[-1]
        return tuple(buf[..., i * d:(i + 1) * d] for i in range(3))
    return tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in arrs)


@pytest.mark.parametrize("name", list(cases.ATTENTION_CASES))
def test_golden_vectors_through_c_abi(name, gpu_device):
    case = cases.ATTENTION_CASES[name]
    Q, K, V = _dev(cases.make_inputs(case), gpu_device)
    if case["layout"] == "qkv":
        assert Q.stride(0) == 3 * case["d_model"]          # the kernel sees the model's strided views
    out = cmhsa.attention_fwd(Q, K, V, case["H"])
    torch.cuda.synchronize()
    z = np.load(cases.golden_path(name))
    err = np.abs(out.cpu().numpy()[z["rows"]] - z["expected"]).max()
    assert err <= (TOL_RAND if case["dist"] == "rand" else TOL), err
    assert out.is_contiguous() and out.shape == Q.shape


def test_reference_sweep_seq_len_1_to_199(gpu_device):
    """Counterpart of reference test_torch_extension.py:11-22: H=12, d_model=768, seq_len 1..199,
    torch.rand inputs, custom backend vs the naive formula -- same call shapes, through the torch op."""
    import causal_multihead_self_attention  # noqa: F401  (registers the op, test_torch_extension.py:3)

    torch.manual_seed(0)
    worst_cpu = worst_same_device = 0.0
    for seq_len in range(1, 200):
        Q, K, V = (torch.rand((1, seq_len, 768), device=gpu_device) for _ in range(3))
        got = scaled_dot_product_attention(CUSTOM, queries=Q, keys=K, values=V, num_heads=12, causal_mask=None)
        assert got.shape == (1, seq_len, 768)
        # (i) against the CPU oracle (the reference formula in torch fp32 on the host)
        want = oracle.naive_attention_torch(Q.cpu(), K.cpu(), V.cpu(), 12)
        worst_cpu = max(worst_cpu, (got.cpu() - want).abs().max().item())
        # (ii) exactly what the reference test compares (test_torch_extension.py:19-22): custom vs the NAIVE backend
        # on the SAME device, same mask construction
        mask = torch.triu(torch.ones(seq_len, seq_len, device=gpu_device), diagonal=1)
        naive = scaled_dot_product_attention(NAIVE, queries=Q, keys=K, values=V, num_heads=12, causal_mask=mask)
        worst_same_device = max(worst_same_device, (got - naive).abs().max().item())
    assert worst_cpu < TOL_REF, worst_cpu
    assert worst_same_device < TOL_REF, worst_same_device


def test_sweep_randn_against_f64_oracle(gpu_device):
    rng = np.random.default_rng(5)
    worst = 0.0
    for T in (1, 2, 3, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 160, 255, 256, 257):
        q, k, v = (rng.standard_normal((T, 768), dtype=np.float32) for _ in range(3))
        out = cmhsa.attention_fwd(*_dev((q, k, v), gpu_device), 12).cpu().numpy()
        worst = max(worst, np.abs(out - oracle.naive_attention_f64(q, k, v, 12)).max())
    assert worst <= TOL, worst


@pytest.mark.parametrize("d_model,H", [(768, 12), (1024, 16), (1280, 20), (1600, 25), (64, 1), (192, 3),
                                        (832, 13), (960, 15), (1088, 17)])   # 13..17 heads: grids around one round of CUs,
                                                                             # where the plan picks a larger chain cap
def test_all_model_widths_t1024_strided(d_model, H, gpu_device):
    """T=1024 (the reference's max context), every GPT-2 width (CMHSA.cu:435-449) plus odd head
    counts, inputs as QKV slices.  Checked on a head subset against the f64 oracle."""
    rng = np.random.default_rng(d_model)
    T = 1024
    buf = rng.standard_normal((T, 3 * d_model), dtype=np.float32)
    q, k, v = buf[:, :d_model], buf[:, d_model:2 * d_model], buf[:, 2 * d_model:]
    out = cmhsa.attention_fwd(*_dev((q, k, v), gpu_device), H).cpu().numpy()
    for h in sorted({0, H // 2, H - 1}):
        sl = slice(64 * h, 64 * h + 64)
        want = oracle.naive_attention_f64(q[:, sl], k[:, sl], v[:, sl], 1)
        assert np.abs(out[:, sl] - want).max() <= TOL


def test_torch_op_equals_c_abi_bitwise(gpu_device):
    import causal_multihead_self_attention  # noqa: F401

    torch.manual_seed(1)
    qkv = torch.randn(300, 3 * 768, device=gpu_device)
    Q, K, V = qkv[:, :768], qkv[:, 768:1536], qkv[:, 1536:]
    a = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_torch(Q=Q, K=K, V=V, num_heads=12)
    b = cmhsa.attention_fwd(Q, K, V, 12, workspace=cmhsa.Workspace(gpu_device))     # the op runs with its stream's workspace
    assert torch.equal(a, b)
    assert a.shape == (300, 768) and a.is_contiguous()


def test_batched_rows_equal_single_calls(gpu_device):
    torch.manual_seed(2)
    B, T, H = 5, 173, 16           # B*H = 80, divisible by 8 -> exercises the XCD-aware block map
    x = torch.randn(B, T, 3 * 64 * H, device=gpu_device)
    Q, K, V = x[..., :1024], x[..., 1024:2048], x[..., 2048:]
    out = cmhsa.attention_fwd(Q, K, V, H)
    again = cmhsa.attention_fwd(Q, K, V, H)
    assert torch.equal(out, again)                                  # same launch shape -> bitwise reproducible
    for b in range(B):
        # the split-KV width (4 or 8 waves) is chosen from the grid size, so a different batch size may
        # sum in a different order: rows agree to rounding, not necessarily to the last bit
        single = cmhsa.attention_fwd(Q[b], K[b], V[b], H)
        assert (out[b] - single).abs().max().item() <= 2e-6
    want = oracle.naive_attention_f64(Q[1].cpu().numpy(), K[1].cpu().numpy(), V[1].cpu().numpy(), H)
    assert np.abs(out[1].cpu().numpy() - want).max() <= TOL


def test_online_softmax_rescale_is_forced(gpu_device):
    """cdna guide rule 26: force the running-max jump.  One late key matches every query strongly,
    so the max jumps by ~hundreds at a chosen unit in some wave's stream; earlier mass must be
    rescaled away exactly once."""
    rng = np.random.default_rng(11)
    T, H = 700, 2
    q = rng.standard_normal((T, 128), dtype=np.float32)
    k = rng.standard_normal((T, 128), dtype=np.float32)
    v = rng.standard_normal((T, 128), dtype=np.float32)
    for spike in (5, 200, 333, 650):
        k2 = k.copy()
        k2[spike, :64] = 6.0 * np.sign(q[spike + 3, :64])        # large score for most later queries
        k2[spike, 64:] = -6.0 * np.sign(q[spike + 3, 64:])
        out = cmhsa.attention_fwd(*_dev((q, k2, v), gpu_device), H).cpu().numpy()
        want = oracle.naive_attention_f64(q, k2, v, H)
        assert np.isfinite(out).all()
        assert np.abs(out - want).max() <= TOL
    # extreme dynamic range: scores up to ~ +-100
    out = cmhsa.attention_fwd(*_dev((4 * q, 4 * k, v), gpu_device), H).cpu().numpy()
    assert np.abs(out - oracle.naive_attention_f64(4 * q, 4 * k, v, H)).max() <= TOL


@pytest.mark.parametrize("slope", [0.6, -0.6, 0.03, 3.0])
def test_speculative_softmax_ramps(gpu_device, slope):
    """The kernel computes P against the OLD running max and only looks at a unit's own max when a row sum leaves
    [.., 2^64) (or is not finite).  Scores that ramp with the key index exercise both sides of that test: a rising ramp
    makes the sums cross the threshold every few units (slope 0.6: ~2^28 per 32-key unit; slope 3: every unit
    overflows to inf), a falling ramp leaves the max stale-high for the whole row (tiny but exact weights), a slow ramp
    stays speculative throughout.  Scores of several hundred carry an absolute fp32 rounding error of ~1e-5 in the dot
    product itself, which ANY fp32 evaluation turns into output error; so the bound is 1e-4 or three times the error
    of the reference formula evaluated in fp32 (the torch oracle), whichever is larger."""
    rng = np.random.default_rng(12)
    T, H = 600, 2
    d = 64 * H
    q = np.ones((T, d), dtype=np.float32) + 0.01 * rng.standard_normal((T, d), dtype=np.float32)
    k = rng.standard_normal((T, d), dtype=np.float32) * 0.05
    k += (slope * 8.0 / 64.0) * np.arange(T, dtype=np.float32)[:, None] / np.float32(1.0)      # q.k/8 ~ slope * key index
    v = rng.standard_normal((T, d), dtype=np.float32)
    out = cmhsa.attention_fwd(*_dev((q, k, v), gpu_device), H).cpu().numpy()
    assert np.isfinite(out).all()
    want = oracle.naive_attention_f64(q, k, v, H)
    ref32 = oracle.naive_attention_torch(torch.from_numpy(q)[None], torch.from_numpy(k)[None], torch.from_numpy(v)[None], H)[0].numpy()
    assert np.abs(out - want).max() <= max(TOL, 3 * np.abs(ref32 - want).max())


def test_every_output_element_is_written(gpu_device):
    # the torch binding hands the kernel torch::empty memory (CMHSA.cu:498): poison it first
    torch.manual_seed(3)
    for T in (1, 33, 100):
        Q, K, V = (torch.rand(T, 768, device=gpu_device) for _ in range(3))
        out = torch.full((T, 768), float("nan"), device=gpu_device)
        cmhsa.attention_fwd(Q, K, V, 12, out=out)
        assert torch.isfinite(out).all()
        assert torch.allclose(out[0], V[0], atol=1e-6)            # row 0 sees key 0 only


def test_module_matches_reference_golden_on_gpu(gpu_device):
    from test_host import _load_module

    mod, x = _load_module(CUSTOM, gpu_device)
    with torch.no_grad():
        y = mod(x).squeeze(0).cpu().numpy()
    z = np.load(cases.golden_path("module_t40_d768"))
    assert np.abs(y - z["expected"]).max() <= TOL
    # and custom vs naive backend on the same device, same weights
    mod_n, _ = _load_module(NAIVE, gpu_device)
    with torch.no_grad():
        assert (mod(x) - mod_n(x)).abs().max().item() <= TOL


def test_error_behaviour_of_the_op(gpu_device):
    import causal_multihead_self_attention  # noqa: F401

    op = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_torch
    x = torch.rand(8, 768, device=gpu_device)
    with pytest.raises(RuntimeError, match="float32"):
        op(Q=x.double(), K=x.double(), V=x.double(), num_heads=12)
    with pytest.raises(RuntimeError, match="Head dimension must be 64"):
        op(Q=x, K=x, V=x, num_heads=6)
    with pytest.raises(RuntimeError, match="evenly divisible"):
        op(Q=x, K=x, V=x, num_heads=5)
    with pytest.raises(RuntimeError, match="same sequence length"):
        op(Q=x, K=x[:4], V=x, num_heads=12)
    with pytest.raises(RuntimeError, match="stride"):
        op(Q=x.t().contiguous().t(), K=x, V=x, num_heads=12)
    empty = op(Q=x[:0], K=x[:0], V=x[:0], num_heads=12)
    assert empty.shape == (0, 768)


def test_non_default_stream(gpu_device):
    torch.manual_seed(4)
    Q, K, V = (torch.randn(257, 768, device=gpu_device) for _ in range(3))
    ref = cmhsa.attention_fwd(Q, K, V, 12)
    s = torch.cuda.Stream(device=gpu_device)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        got = cmhsa.attention_fwd(Q, K, V, 12)
    s.synchronize()
    assert torch.equal(ref, got)


def test_full_size_properties_config5_shape(gpu_device):
    """BASELINE.json config 5 geometry per GPU under 8-way head sharding is (B=32, T=4096, H=2); here a
    (B=4, T=4096, H=16) slab keeps the test short.  Size-independent checks: constant-V invariance,
    row 0 == V[0], linearity in V, and f64-oracle agreement on sampled (batch, head) pairs."""
    torch.manual_seed(5)
    B, T, H = 4, 4096, 16
    d = 64 * H
    Q = torch.randn(B, T, d, device=gpu_device)
    K = torch.randn(B, T, d, device=gpu_device)
    V1 = torch.randn(B, T, d, device=gpu_device)
    V2 = torch.randn(B, T, d, device=gpu_device)
    o1 = cmhsa.attention_fwd(Q, K, V1, H)
    o2 = cmhsa.attention_fwd(Q, K, V2, H)
    o12 = cmhsa.attention_fwd(Q, K, 0.5 * V1 - 2.0 * V2, H)
    assert torch.isfinite(o1).all()
    assert (o12 - (0.5 * o1 - 2.0 * o2)).abs().max().item() <= TOL              # linear in V
    assert torch.allclose(o1[:, 0], V1[:, 0], atol=1e-6)                          # first row
    const = torch.arange(d, device=gpu_device, dtype=torch.float32).expand(B, T, d).contiguous()
    oc = cmhsa.attention_fwd(Q, K, const, H)
    assert (oc - const).abs().max().item() <= 5e-6 * d                            # rows of softmax sum to 1 (relative 5e-6 of max |V|)
    for b, h in ((0, 0), (3, 15), (2, 7)):
        sl = slice(64 * h, 64 * h + 64)
        want = oracle.naive_attention_f64(Q[b, :, sl].cpu().numpy(), K[b, :, sl].cpu().numpy(),
                                          V1[b, :, sl].cpu().numpy(), 1)
        assert np.abs(o1[b, :, sl].cpu().numpy() - want).max() <= TOL


@pytest.mark.parametrize("B,H", [(32, 2), (32, 16)])
def test_config5_literal_shapes(B, H, gpu_device):
    """BASELINE.json configs[4] at its literal sizes: one GPU's head shard (B=32, H=2, T=4096: a column slab of Q, K, V with the
    whole model's row stride, 3 * 1024 floats) and the whole config on one GPU (B=32, H=16: the N = 1 point).  Every
    element written (NaN-filled output), the call bit-reproducible, with the workspace and without it, and sampled
    (batch, head) pairs against the f64 oracle."""
    torch.manual_seed(500 + H)
    T, d_model = 4096, 1024
    d = 64 * H
    qkv = torch.randn(B, T, 3 * d_model, device=gpu_device)          # the model-wide packed activation
    q, k, v = (qkv[..., i * d_model:i * d_model + d] for i in range(3))   # this GPU's heads: the first H of 16
    assert q.stride() == (T * 3 * d_model, 3 * d_model, 1)
    out = torch.full((B, T, d), float("nan"), device=gpu_device)
    cmhsa.attention_fwd(q, k, v, H, out=out, workspace=cmhsa.Workspace(gpu_device))
    assert torch.isfinite(out).all()
    again = cmhsa.attention_fwd(q, k, v, H, workspace=cmhsa.Workspace(gpu_device))
    assert torch.equal(out, again)
    plain = cmhsa.attention_fwd(q, k, v, H)
    assert torch.equal(out, plain)                                  # grids of many rounds: one plan with or without a workspace
    for b, h in ((0, 0), (B - 1, H - 1), (B // 2, H // 2)):
        sl = slice(64 * h, 64 * h + 64)
        want = oracle.naive_attention_f64(q[b, :, sl].cpu().numpy(), k[b, :, sl].cpu().numpy(), v[b, :, sl].cpu().numpy(), 1)
        assert np.abs(out[b, :, sl].cpu().numpy() - want).max() <= TOL, (b, h)
    assert torch.allclose(out[:, 0], v[:, 0], atol=1e-6)            # row 0 attends to key 0 only


@pytest.mark.parametrize("name", [n for n, c in cases.ATTENTION_CASES.items() if c["layout"] == "qkv"])
def test_packed_qkv_entry_point_against_reference_goldens(name, gpu_device):
    """SURVEY 8(f) row 2 / VERDICT f2: the `layout="qkv"` goldens are exactly the packed (T, 3*d_model) buffer the
    fused QKV layer produces (causal_self_attention.py:224-227); fed WHOLE to cmhsa_fp32_fwd_packed_qkv (C ABI and
    torch op) the result must match the outputs the reference module produced (tests/golden/*.npz)."""
    import causal_multihead_self_attention  # noqa: F401

    case = cases.ATTENTION_CASES[name]
    q, k, v = cases.make_inputs(case)
    packed = torch.from_numpy(q.base).to(gpu_device)                 # the one (T, 3*d_model) buffer behind the three views
    assert packed.shape == (case["T"], 3 * case["d_model"])
    z = np.load(cases.golden_path(name))
    tol = TOL_RAND if case["dist"] == "rand" else TOL
    via_abi = cmhsa.attention_fwd_packed_qkv(packed, case["H"])
    assert np.abs(via_abi.cpu().numpy()[z["rows"]] - z["expected"]).max() <= tol
    via_op = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_packed_qkv(QKV=packed, num_heads=case["H"])
    assert np.abs(via_op.cpu().numpy()[z["rows"]] - z["expected"]).max() <= tol
    # and against the f64 oracle on every row (the goldens of the T=1024 case store every 41st row only)
    want = oracle.naive_attention_f64(q, k, v, case["H"])
    assert np.abs(via_abi.cpu().numpy() - want).max() <= TOL


@pytest.mark.parametrize("B", [8, 64])
def test_config4_attention_shape_gpt2_xl(B, gpu_device):
    """BASELINE.json configs[3]: GPT2-XL (H=25, d_model=1600, gpt2.py:70), 64 prompts (8 per GPU at 8 GPUs), T=512,
    Q/K/V as slices of the packed QKV activation.  B*H = 1600 (or 200) drives the XCD-pinned grid map
    (gridDim.z = B*H/8); sampled (batch, head) pairs against the f64 oracle, every row written, batch rows of equal
    inputs bit-identical."""
    torch.manual_seed(40 + B)
    T, H = 512, 25
    d = 64 * H
    qkv = torch.randn(B, T, 3 * d, device=gpu_device)
    qkv[B - 1] = qkv[0]                                             # identical prompts must give identical rows
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    out = torch.full((B, T, d), float("nan"), device=gpu_device)
    cmhsa.attention_fwd(q, k, v, H, out=out)
    assert torch.isfinite(out).all()
    assert torch.equal(out[0], out[B - 1])
    assert torch.equal(cmhsa.attention_fwd_packed_qkv(qkv, H), out)
    qc, kc, vc, oc = q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), out.cpu().numpy()
    for b, h in ((0, 0), (B - 1, 24), (B // 2, 12), (1, 7), (B - 2, 17)):
        sl = slice(64 * h, 64 * h + 64)
        want = oracle.naive_attention_f64(qc[b, :, sl], kc[b, :, sl], vc[b, :, sl], 1)
        assert np.abs(oc[b, :, sl] - want).max() <= TOL, (b, h)


PAIRS, LONGEST_FIRST = "one wave per pair of query blocks", "one wave per query block, longest first"


@pytest.mark.parametrize("B,H,T,family", [(16, 16, 512, PAIRS), (16, 16, 500, PAIRS), (16, 16, 480, PAIRS), (16, 16, 449, PAIRS),
                                          (4, 16, 2048, PAIRS), (85, 3, 490, PAIRS), (5, 12, 1001, PAIRS),
                                          (16, 12, 500, LONGEST_FIRST), (8, 25, 512, LONGEST_FIRST), (7, 25, 600, LONGEST_FIRST),
                                          (21, 9, 1000, LONGEST_FIRST), (16, 12, 1024, LONGEST_FIRST)])
def test_one_wave_workgroup_plans(B, H, T, family, gpu_device):
    """Larger grids get ONE-WAVE workgroups: a wave runs whole query blocks (no split-KV, no LDS merge) and writes O
    straight from its accumulators -- either the pair n and nqb+1-n one after the other (equal jobs; the prefetch stream
    runs on across the two blocks), or one block per wave with the grid ordered longest block first across all (b,h)s
    (grid x = B*H padded to a multiple of 8: the padding workgroups must exit).  Exact and ragged lengths, an odd number
    of query blocks (the middle one has no partner), B*H not a multiple of 8, under half a round of jobs: every element
    written, the whole output against the oracle's formula in float64, batch rows of equal inputs bit-identical, strided
    (packed-QKV) inputs, bit-reproducible."""
    assert cmhsa.plan_description(B, T, H).startswith(family + ":"), cmhsa.plan_description(B, T, H)
    torch.manual_seed(B * 1000 + T)
    d = 64 * H
    qkv = torch.randn(B, T, 3 * d, device=gpu_device)
    qkv[B - 1] = qkv[0]
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    out = torch.full((B, T, d), float("nan"), device=gpu_device)
    cmhsa.attention_fwd(q, k, v, H, out=out)
    assert torch.isfinite(out).all()
    assert torch.equal(out[0], out[B - 1])
    want = oracle.naive_attention_torch(q.double(), k.double(), v.double(), H)
    assert (out.double() - want).abs().max().item() <= TOL
    assert torch.equal(cmhsa.attention_fwd(q, k, v, H), out)          # and again: bit-reproducible


LISTS = "job lists, one wave per list"


@pytest.mark.parametrize("B,H,T", [(32, 25, 512), (37, 25, 511), (64, 25, 256), (64, 25, 500)])
def test_job_list_plans(B, H, T, gpu_device):
    """Many (b,h)s of short sequences (config 4's attention shapes) run as ONE round of one-wave workgroups that each work
    through a list of whole query blocks of several (b,h)s (a period of 25-29 (b,h)s dealt to 32 or 64 lists; blocks of
    up to nqb / 2 units taken from the diagonal down, wherever they are).  (b,h) counts that are and are not a multiple of the period
    (37 * 25 = 31 periods of 29 + 26), exact and ragged lengths: every element written, the whole output against the oracle's
    formula in float64, batch rows of equal inputs bit-identical, packed-QKV strides, bit-reproducible."""
    assert cmhsa.plan_description(B, T, H).startswith(LISTS + ":"), cmhsa.plan_description(B, T, H)
    torch.manual_seed(B * 1000 + T)
    d = 64 * H
    qkv = torch.randn(B, T, 3 * d, device=gpu_device)
    qkv[:, T // 2:, d:2 * d] *= 3.0                                # later keys score higher: the running max moves inside a block
    qkv[B - 1] = qkv[0]
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    out = torch.full((B, T, d), float("nan"), device=gpu_device)
    cmhsa.attention_fwd(q, k, v, H, out=out)
    assert torch.isfinite(out).all()
    assert torch.equal(out[0], out[B - 1])
    for b0 in range(0, B, 8):                                     # (float64 scores of 8 batch rows at a time: memory)
        sl = slice(b0, min(B, b0 + 8))
        want = oracle.naive_attention_torch(q[sl].double(), k[sl].double(), v[sl].double(), H)
        assert (out[sl].double() - want).abs().max().item() <= TOL, b0
    assert torch.equal(cmhsa.attention_fwd(q, k, v, H), out)          # and again: bit-reproducible
    assert torch.equal(cmhsa.attention_fwd_packed_qkv(qkv, H), out)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"CMHSA_LIST": "1"}, {"CMHSA_LIST": "1", "CMHSA_LIST_REV": "0"}, {"CMHSA_LIST": "1", "CMHSA_LIST_SCHED": "0"},
                                 {"CMHSA_LIST": "1", "CMHSA_LIST_P": "7"}, {"CMHSA_LIST": "0"}])
def test_forced_job_lists_in_a_child_process(env):
    """The job-list family wherever it applies (plan-forcing test build, as in test_forced_plans_in_a_child_process): both
    schedules, with and without descending short blocks, a forced period length, and switched off; (b,h) counts that leave a short
    last period, one-unit blocks (two levels of look-ahead across block boundaries), a single (b,h) per period."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path[:0] = [%r, %r]
import cmhsa
cmhsa.LIB_PATH = %r
from oracle import naive_attention as oracle
rng = np.random.default_rng(78)
worst, on_lists = 0.0, 0
for B, T, H in ((100, 300, 7), (333, 200, 3), (41, 777, 16), (16, 512, 25), (1000, 480, 1), (64, 65, 25), (64, 128, 25), (9, 1024, 12), (203, 100, 5)):
    d = 64 * H
    qkv = rng.standard_normal((B, T, 3 * d), dtype=np.float32)
    qkv[:, T // 2:, d:2 * d] *= 4.0
    t = torch.from_numpy(qkv).cuda()
    on_lists += cmhsa.plan_description(B, T, H).startswith("job lists")
    out = torch.full((B, T, d), float("nan"), device="cuda")
    cmhsa.attention_fwd(t[..., :d], t[..., d:2 * d], t[..., 2 * d:], H, out=out)
    assert torch.isfinite(out).all(), (B, T, H)           # every element written
    split = lambda x: x.reshape(B, T, H, 64).transpose(1, 2)
    with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.MATH):   # every (b,h), fp32: a wrong base or a dropped job shows
        ref = torch.nn.functional.scaled_dot_product_attention(split(t[..., :d]), split(t[..., d:2 * d]), split(t[..., 2 * d:]), is_causal=True)
    assert (out - ref.transpose(1, 2).reshape(B, T, d)).abs().max().item() <= 3e-4, (B, T, H)
    out = out.cpu().numpy()
    for b in sorted({0, B // 2, B - 1}):                  # and sampled (b,h)s against the f64 oracle at the parity bound
        for h in sorted({0, H - 1}):
            sl = slice(64 * h, 64 * h + 64)
            want = oracle.naive_attention_f64(qkv[b, :, sl], qkv[b, :, d + 64 * h:d + 64 * h + 64], qkv[b, :, 2 * d + 64 * h:2 * d + 64 * h + 64], 1)
            worst = max(worst, float(np.abs(out[b, :, sl] - want).max()))
print("WORST", worst, "ON_LISTS", on_lists)
assert worst <= 1e-4
""" % (root, os.path.join(root, "llm-inference_amd"), os.path.join(root, "tests", "helpers", "libcmhsa_hip_planenv.so"))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    n = int(r.stdout.split("ON_LISTS")[1].split()[0])
    assert (n >= 5) if env["CMHSA_LIST"] == "1" else (n == 0), r.stdout[-500:]


@pytest.mark.gpu
def test_packed_qkv_entry_point_equals_the_three_view_call(gpu_device):
    """one packed (.., T, 3*d_model) buffer in -- same kernel, so the same bits as slicing it into
    the three views the reference hands over (causal_self_attention.py:224-227); C ABI, torch op and module flag."""
    import causal_multihead_self_attention  # noqa: F401
    from causal_self_attention import CausalSelfAttention, ScaledDotProductAttentionBackend as Backend

    torch.manual_seed(21)
    for B, T, H in ((1, 1, 12), (1, 199, 12), (1, 1024, 12), (3, 130, 25), (2, 64, 16)):
        d = 64 * H
        qkv = torch.randn(B, T, 3 * d, device=gpu_device)
        want = cmhsa.attention_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H)
        assert torch.equal(cmhsa.attention_fwd_packed_qkv(qkv, H), want)
        ws = cmhsa.Workspace(gpu_device)                  # with a workspace both forms may split query blocks: same plan, same bits
        want_ws = cmhsa.attention_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, workspace=ws)
        assert torch.equal(cmhsa.attention_fwd_packed_qkv(qkv, H, workspace=ws), want_ws)
        assert (want_ws - want).abs().max().item() <= 5e-6
        # the torch ops run with their stream's workspace
        op = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_packed_qkv
        assert torch.equal(op(QKV=qkv, num_heads=H), want_ws)
        if B == 1:
            assert torch.equal(op(QKV=qkv[0], num_heads=H), want_ws[0])
    # padded rows (row stride > 3*d_model) are fine, too-narrow / mis-shaped inputs raise
    wide = torch.randn(50, 3 * 768 + 64, device=gpu_device)
    got = cmhsa.attention_fwd_packed_qkv(wide[:, :3 * 768], 12)
    assert torch.equal(got, cmhsa.attention_fwd(wide[:, :768], wide[:, 768:1536], wide[:, 1536:2304], 12))
    with pytest.raises(RuntimeError):
        op(QKV=torch.randn(8, 3 * 768 + 4, device=gpu_device), num_heads=12)
    with pytest.raises(RuntimeError):
        op(QKV=torch.randn(8, 3 * 96, device=gpu_device), num_heads=3)           # head_dim 32
    torch.manual_seed(22)
    a = CausalSelfAttention(768, 768, 256, 0.1, 12, True, Backend.CUSTOM_CUDA).to(gpu_device).eval()
    b = CausalSelfAttention(768, 768, 256, 0.1, 12, True, Backend.CUSTOM_CUDA, packed_qkv=True).to(gpu_device).eval()
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 200, 768, device=gpu_device)
    with torch.no_grad():
        assert torch.equal(a(x), b(x))


@pytest.mark.gpu
@pytest.mark.parametrize("T,H", [(16384, 1), (12345, 3), (33000, 1)])
def test_long_sequences_spot_rows_against_f64(gpu_device, T, H):
    """Long, ragged sequences (hundreds to a thousand query blocks: every branch of the packing plan, both workgroup
    shapes): spot rows against an fp64 evaluation of the reference formula on the GPU's inputs, first row = V[0],
    constant-V invariance, full write of the output."""
    torch.manual_seed(31)
    d = 64 * H
    qkv = torch.randn(T, 3 * d, device=gpu_device)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    out = torch.full((T, d), float("nan"), device=gpu_device)
    cmhsa.attention_fwd(q, k, v, H, out=out)
    assert not torch.isnan(out).any()
    assert torch.allclose(out[0], v[0], atol=1e-6)
    rows = [1, 31, 32, 33, 1023, 1024, T // 2, T - 33, T - 32, T - 1]
    q64, k64, v64 = q.double(), k.double(), v.double()
    for h in range(H):
        sl = slice(64 * h, 64 * h + 64)
        for r in rows:
            w = torch.softmax((k64[:r + 1, sl] @ q64[r, sl]) / 8.0, dim=0)
            want = w @ v64[:r + 1, sl]
            assert (out[r, sl].double() - want).abs().max().item() <= 1e-4
    ones = torch.full_like(v, 0.75)
    const = cmhsa.attention_fwd(q, k, ones, H)
    assert (const - 0.75).abs().max().item() <= 5e-6 * 64


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"CMHSA_TABLE": "1", "CMHSA_WAVES": "8"}, {"CMHSA_TABLE": "1", "CMHSA_WAVES": "4"},
                                 {"CMHSA_TABLE": "1", "CMHSA_WAVES": "8", "CMHSA_CHAIN": "9"},
                                 {"CMHSA_TABLE": "0", "CMHSA_WAVES": "8", "CMHSA_CHAIN": "7"},
                                 {"CMHSA_WAVES": "1"}, {"CMHSA_WAVES": "1", "CMHSA_LPT": "16"}, {"CMHSA_WAVES": "1", "CMHSA_LPT": "1000000"},
                                 # the 16-key-tile family (csrc/cmhsa_fwd16.hip) wherever it fits, in each of its shapes: 16- / 32-row
                                 # blocks, one / two waves per SIMD, both workgroup -> (b,h) maps; and switched off altogether
                                 {"CMHSA_T16": "1", "CMHSA_T16_SUB": "1", "CMHSA_T16_WAVES": "4"},
                                 {"CMHSA_T16": "1", "CMHSA_T16_SUB": "1", "CMHSA_T16_WAVES": "8", "CMHSA_T16_CONTIG": "1"},
                                 {"CMHSA_T16": "1", "CMHSA_T16_SUB": "2", "CMHSA_T16_WAVES": "4", "CMHSA_T16_CONTIG": "0"},
                                 {"CMHSA_T16": "1", "CMHSA_T16_SUB": "2", "CMHSA_T16_WAVES": "8"},
                                 {"CMHSA_T16": "0"}])
def test_forced_plans_in_a_child_process(env):
    """The plan (workgroup shape, chain cap, analytic or table packing with up to four query blocks per workgroup) is
    chosen per shape by a cost model.  The shipped library reads no environment variable; the TEST build of the same
    sources (tests/helpers/libcmhsa_hip_planenv.so, -DCMHSA_PLAN_ENV) does, once per process, so each forced combination
    runs in a child that points the ctypes wrapper at that build: ragged and exact lengths, one to many heads, against
    the f64 oracle at 1e-4."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path[:0] = [%r, %r]
import cmhsa
cmhsa.LIB_PATH = %r          # the plan-forcing test build of the same kernel sources
from oracle import naive_attention as oracle
rng = np.random.default_rng(77)
worst = 0.0
for B, T, H in ((1, 1, 1), (1, 16, 12), (1, 17, 12), (1, 31, 2), (1, 33, 3), (1, 100, 25), (2, 257, 4), (1, 500, 12), (1, 1000, 5), (1, 1024, 20),
                (3, 700, 12), (1, 2049, 2), (1, 4064, 3), (1, 8100, 1)):
    qkv = rng.standard_normal((B, T, 3 * 64 * H), dtype=np.float32)
    if T > 64:
        qkv[:, T // 2:, 64 * H:2 * 64 * H] *= 4.0          # later keys score higher: the running max moves inside and across jobs
    t = torch.from_numpy(qkv).cuda()
    d = 64 * H
    out = torch.full((B, T, d), float("nan"), device="cuda")
    cmhsa.attention_fwd(t[..., :d], t[..., d:2 * d], t[..., 2 * d:], H, out=out)
    out = out.cpu().numpy()
    assert np.isfinite(out).all(), (B, T, H)              # every element written
    for b in range(B):
        for h in sorted({0, H - 1}):
            sl = slice(64 * h, 64 * h + 64)
            want = oracle.naive_attention_f64(qkv[b, :, sl], qkv[b, :, d + 64 * h:d + 64 * h + 64], qkv[b, :, 2 * d + 64 * h:2 * d + 64 * h + 64], 1)
            worst = max(worst, float(np.abs(out[b, :, sl] - want).max()))
print("WORST", worst, cmhsa.plan_description(1, 500, 12))
assert worst <= 1e-4
""" % (root, os.path.join(root, "llm-inference_amd"), os.path.join(root, "tests", "helpers", "libcmhsa_hip_planenv.so"))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "WORST" in r.stdout
    if "CMHSA_T16" in env:                                # the forced family really ran (B=1, T=500, H=12 fits it)
        assert ("16-key tiles" in r.stdout) == (env["CMHSA_T16"] == "1"), r.stdout[-500:]
        if env["CMHSA_T16"] == "1":
            assert ("%s-row blocks" % (16 * int(env["CMHSA_T16_SUB"]))) in r.stdout and ("%s-wave" % env["CMHSA_T16_WAVES"]) in r.stdout, r.stdout[-500:]


def test_random_shapes_fuzz(gpu_device):
    """Forty seeded random (B, H, T) shapes -- small to mid-size, so every branch of the plan choice (4/8 waves, chain
    caps above the minimum, analytic and table packing) and ragged tails get hit -- against the oracle's torch
    formula evaluated in float64 on the same inputs (run on the GPU only to keep the test fast: it is the oracle's
    code, oracle/naive_attention.py, not the product's)."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for _ in range(40):
        B = int(rng.integers(1, 5)); H = int(rng.integers(1, 27)); T = int(rng.integers(1, 1400))
        d = 64 * H
        g = torch.Generator(device="cpu").manual_seed(int(rng.integers(1 << 30)))
        qkv = torch.randn((B, T, 3 * d), generator=g).to(gpu_device)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        out = cmhsa.attention_fwd(q, k, v, H)
        want = oracle.naive_attention_torch(q.double(), k.double(), v.double(), H)
        err = (out.double() - want).abs().max().item()
        worst = max(worst, err)
        assert err <= TOL, (B, H, T, err)
    assert worst <= 1e-5, worst          # randn inputs: observed ~1e-6


@pytest.mark.parametrize("B,H,T", [(1, 16, 2048), (1, 12, 3072), (1, 25, 2048), (1, 12, 4000), (2, 12, 2048), (1, 16, 3000), (1, 12, 4096),
                                   (1, 12, 6144), (1, 12, 4097), (3, 12, 3000)])
def test_single_long_prompts(B, H, T, gpu_device):
    """Single long prompts (packed tables with eight segments per workgroup where they fit; beyond, the 16-key-tile kernel with
    32-row blocks -- job lists cut at tile granularity, job words that count PAIRS of tiles past T = 4080 -- or wave pairs):
    the whole output against the oracle's formula in float64, every element written, strided (packed-QKV) inputs, with the
    workspace and without it."""
    torch.manual_seed(T + H)
    d = 64 * H
    qkv = torch.randn(B, T, 3 * d, device=gpu_device)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    want = oracle.naive_attention_torch(q.double(), k.double(), v.double(), H)
    ws = cmhsa.Workspace(gpu_device)
    for workspace in (None, ws):
        out = torch.full((B, T, d), float("nan"), device=gpu_device)
        cmhsa.attention_fwd(q, k, v, H, out=out, workspace=workspace)
        assert torch.isfinite(out).all()
        assert (out.double() - want).abs().max().item() <= 1e-5, (B, H, T, workspace is not None)
    plan = cmhsa.plan_dump(B, T, H, True)
    if (B, H, T) == (1, 16, 2048):
        assert plan["family"] == "units32" and plan["use_table"] and plan["segs"] == 8, plan
    if T >= 3000 and H == 12:
        assert plan["family"] == "tiles16" and plan["sub"] == 2 and plan["waves"] == 8 and plan["granule"] == (2 if T > 4080 else 1), plan


def test_random_larger_grids_fuzz(gpu_device):
    """Thirty seeded random shapes with 1 500 to 40 000 (batch, head, query block) triples -- the range where the planner
    moves between the analytic plan, wave pairs and longest-first one-wave workgroups -- ragged lengths, B*H mostly not
    a multiple of 8; whole output against the oracle's formula in float64, and the plan family recorded so that the test
    fails if the sweep stops reaching the one-wave families."""
    rng = np.random.default_rng(77)
    families = set()
    done = 0
    while done < 30:
        B = int(rng.integers(1, 65)); H = int(rng.integers(1, 27)); T = int(rng.integers(40, 1300))
        nqb = (T + 31) // 32
        if not (1500 <= B * H * nqb <= 40000) or B * H * T * T > 6e8:
            continue
        done += 1
        families.add(cmhsa.plan_description(B, T, H).split(":")[0])
        d = 64 * H
        g = torch.Generator(device="cpu").manual_seed(int(rng.integers(1 << 30)))
        qkv = torch.randn((B, T, 3 * d), generator=g).to(gpu_device)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        out = torch.full((B, T, d), float("nan"), device=gpu_device)
        cmhsa.attention_fwd(q, k, v, H, out=out)
        want = oracle.naive_attention_torch(q.double(), k.double(), v.double(), H)
        err = (out.double() - want).abs().max().item()
        assert err <= 1e-5, (B, H, T, err, cmhsa.plan_description(B, T, H))
        del qkv, out, want
    assert PAIRS in families and LONGEST_FIRST in families, families
