"""ctypes view of the C ABI in include/cmhsa.h, include/gpt2_gelu.h and include/gpt2_layernorm.h (libcmhsa_hip.so).

This is the host-side plumbing only: device memory and streams come from PyTorch-ROCm, the
arithmetic is entirely in the hand-written gfx950 kernels.  There is NO fallback: if the library
is missing or no GPU is present, the calls raise.

  attention_fwd / attention_fwd_raw      cmhsa_fp32_fwd            full-sequence causal attention (the hot path)
  attention_fwd(..., workspace=ws)       cmhsa_fp32_fwd_ws         the same with a device workspace (class Workspace): query
                                                                   blocks may then be split across workgroups
  attention_fwd_packed_qkv               cmhsa_fp32_fwd_packed_qkv the same over one packed (.., T, 3*d_model) buffer
  decode_step                            cmhsa_fp32_decode_step    one-token step over a KV cache (opt-in)
  gelu_tanh_fwd                          gpt2_gelu_tanh_f32_fwd    fused tanh-GELU
  sustained_shader_clock_mhz             cmhsa_probe_shader_clock  measurement aid

The torch ops of the sibling extension module `causal_multihead_self_attention` (importing it registers them,
as the reference does at causal_self_attention.py:60-63) go through the same entry points; tests call both.
"""
from __future__ import annotations

import ctypes
import os

import torch  # must be imported BEFORE the library: both must share torch's libamdhip64

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcmhsa_hip.so")

# every symbol include/*.h declares; tests/test_abi.py checks the headers against this list
EXPORTED_SYMBOLS = (
    "cmhsa_status_string",
    "cmhsa_version",
    "cmhsa_fp32_fwd",
    "cmhsa_fp32_fwd_ws",
    "cmhsa_fp32_fwd_noalloc",
    "cmhsa_workspace_bytes",
    "cmhsa_workspace_init",
    "cmhsa_fp32_fwd_workspace_bytes",
    "cmhsa_plan_describe",
    "cmhsa_plan_dump",
    "cmhsa_fp32_fwd_2d",
    "cmhsa_fp32_fwd_packed_qkv",
    "cmhsa_fp32_decode_step",
    "cmhsa_decode_algorithmic_bytes",
    "cmhsa_algorithmic_flops",
    "cmhsa_probe_shader_clock",
    "gpt2_gelu_tanh_f32_fwd",        # include/gpt2_gelu.h
    "gpt2_gelu_algorithmic_bytes",
    "gpt2_add_layernorm_f32_fwd",    # include/gpt2_layernorm.h
    "gpt2_add_layernorm_algorithmic_bytes",
    "gpt2_embed_f32_fwd",            # include/gpt2_step.h
    "gpt2_greedy_step",
    "gpt2_linear_rows_f32_fwd",      # include/gpt2_linear.h
    "gpt2_ln_linear_rows_f32_fwd",
    "gpt2_linear_rows_algorithmic_bytes",
)

CMHSA_OK = 0
CMHSA_ERR_NULL_POINTER = -1
CMHSA_ERR_BAD_SHAPE = -2
CMHSA_ERR_HEAD_DIM = -3
CMHSA_ERR_ALIGNMENT = -4
CMHSA_ERR_TOO_LARGE = -5
CMHSA_ERR_WORKSPACE = -6
NOALLOC_SCRATCH_BYTES = 1 << 20          # CMHSA_NOALLOC_SCRATCH_BYTES (include/cmhsa.h)

_lib = None


class CmhsaError(RuntimeError):
    def __init__(self, status: int, text: str):
        super().__init__(f"cmhsa status {status}: {text}")
        self.status = status


def load_library() -> ctypes.CDLL:
    """dlopen libcmhsa_hip.so (once) and declare the prototypes of include/cmhsa.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python llm-inference_amd/build.py` "
            "(there is no CPU or PyTorch fallback for the HIP attention path)")
    lib = ctypes.CDLL(LIB_PATH)
    i64, i32, vp, sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    lib.cmhsa_status_string.restype = ctypes.c_char_p
    lib.cmhsa_status_string.argtypes = [i32]
    lib.cmhsa_version.restype = i32
    lib.cmhsa_version.argtypes = []
    lib.cmhsa_fp32_fwd.restype = i32
    lib.cmhsa_fp32_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, vp]
    lib.cmhsa_fp32_fwd_ws.restype = i32
    lib.cmhsa_fp32_fwd_ws.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, vp, sz, vp]
    lib.cmhsa_fp32_fwd_noalloc.restype = i32
    lib.cmhsa_fp32_fwd_noalloc.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, vp]
    lib.cmhsa_workspace_bytes.restype = sz
    lib.cmhsa_workspace_bytes.argtypes = []
    lib.cmhsa_workspace_init.restype = i32
    lib.cmhsa_workspace_init.argtypes = [vp, sz, vp]
    lib.cmhsa_fp32_fwd_workspace_bytes.restype = sz
    lib.cmhsa_fp32_fwd_workspace_bytes.argtypes = [i32, i32, i32]
    lib.cmhsa_plan_describe.restype = i32
    lib.cmhsa_plan_describe.argtypes = [i32, i32, i32, i32, ctypes.c_char_p, i32]
    lib.cmhsa_plan_dump.restype = i32
    lib.cmhsa_plan_dump.argtypes = [i32, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int), i32]
    lib.cmhsa_fp32_fwd_2d.restype = i32
    lib.cmhsa_fp32_fwd_2d.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, vp]
    lib.cmhsa_fp32_fwd_packed_qkv.restype = i32
    lib.cmhsa_fp32_fwd_packed_qkv.argtypes = [vp, vp, i64, i64, i64, i64, i32, i32, i32, i32, vp, sz, vp]
    lib.cmhsa_fp32_decode_step.restype = i32
    lib.cmhsa_fp32_decode_step.argtypes = [vp, i64, vp, vp, i64, i64, vp, i64, vp, i32, i32, i32, i32, vp]
    lib.cmhsa_decode_algorithmic_bytes.restype = ctypes.c_double
    lib.cmhsa_decode_algorithmic_bytes.argtypes = [i32, i32, i32, i32]
    lib.cmhsa_probe_shader_clock.restype = i32
    lib.cmhsa_probe_shader_clock.argtypes = [vp, i32, vp]
    lib.cmhsa_algorithmic_flops.restype = ctypes.c_double
    lib.cmhsa_algorithmic_flops.argtypes = [i32, i32, i32, i32]
    lib.gpt2_gelu_tanh_f32_fwd.restype = i32
    lib.gpt2_gelu_tanh_f32_fwd.argtypes = [vp, vp, i64, vp]
    lib.gpt2_gelu_algorithmic_bytes.restype = ctypes.c_double
    lib.gpt2_gelu_algorithmic_bytes.argtypes = [i64]
    lib.gpt2_add_layernorm_f32_fwd.restype = i32
    lib.gpt2_add_layernorm_f32_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i64, i64, ctypes.c_float, vp]
    lib.gpt2_add_layernorm_algorithmic_bytes.restype = ctypes.c_double
    lib.gpt2_add_layernorm_algorithmic_bytes.argtypes = [i64, i32, i32]
    lib.gpt2_embed_f32_fwd.restype = i32
    lib.gpt2_embed_f32_fwd.argtypes = [vp, i64, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.gpt2_greedy_step.restype = i32
    lib.gpt2_greedy_step.argtypes = [vp, i64, vp, vp, i64, vp, i32, i32, i32, i64, vp]
    lib.gpt2_linear_rows_f32_fwd.restype = i32
    lib.gpt2_linear_rows_f32_fwd.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.gpt2_ln_linear_rows_f32_fwd.restype = i32
    lib.gpt2_ln_linear_rows_f32_fwd.argtypes = [vp, i64, vp, i64, vp, vp, ctypes.c_float, vp, vp, i64, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.gpt2_linear_rows_algorithmic_bytes.restype = ctypes.c_double
    lib.gpt2_linear_rows_algorithmic_bytes.argtypes = [i32, i32, i32, i32]
    _lib = lib
    return lib


def status_string(status: int) -> str:
    return load_library().cmhsa_status_string(status).decode()


def algorithmic_flops(batch: int, seq_len: int, num_heads: int, head_dim: int = 64) -> float:
    """F = 2*B*H*d*T*(T+1); pure host arithmetic, no GPU needed."""
    return float(load_library().cmhsa_algorithmic_flops(batch, seq_len, num_heads, head_dim))


def plan_dump(batch: int, seq_len: int, num_heads: int, with_workspace: bool = False, cus: int = 0) -> dict:
    """The plan of a shape as data (cmhsa_plan_dump; no GPU needed when `cus` is given): header fields and, per workgroup
    of a (b,h), its segments as (n, first_unit, unit_count, wave_mask, part_word) tuples (absent segments dropped)."""
    lib = load_library()
    need = lib.cmhsa_plan_dump(batch, seq_len, num_heads, 1 if with_workspace else 0, cus, None, 0)
    if need <= 0:
        raise CmhsaError(need, status_string(need))
    buf = (ctypes.c_int * need)()
    got = lib.cmhsa_plan_dump(batch, seq_len, num_heads, 1 if with_workspace else 0, cus, buf, need)
    if got != need:
        raise CmhsaError(got, status_string(got))
    names = ("waves", "segs", "wgs_per_bh", "grid_x", "grid_y", "grid_z", "parts_per_bh", "splits_per_bh", "simd_cap", "nqb",
             "use_table", "longest_first", "chain", "n_paired", "n_alone", "workspace_256")
    plan = dict(zip(names, buf[:16]))
    if plan["use_table"] == 2:
        # the 16-row-block family (csrc/cmhsa_tiles16.h): "segs" = job words per wave, "nqb" = 16-row blocks per (b,h),
        # "simd_cap" = tiles on the busiest SIMD; per workgroup its 4 or 8 job lists in run order (with 8 waves the lists
        # 2s and 2s + 1 share SIMD s), per list its jobs as (block, first_tile, tiles, rank, pieces); "merges" = per
        # workgroup, the barrier flag of each list's first word
        J, W = plan["segs"], plan["waves"]
        plan.update(family="tiles16", jobs_per_wave=J, blocks=plan["nqb"], tiles_per_simd=plan["simd_cap"], sub=plan["chain"],
                    contig=bool(plan["n_paired"]))
        words = [[[buf[16 + (j * W + w) * J + i] & 0xffffffff for i in range(J)] for w in range(W)] for j in range(plan["wgs_per_bh"])]
        gran = max(1, plan["n_alone"])                     # tiles per count of a job word's first-tile / tiles fields
        plan["granule"] = gran
        plan["workgroups"] = [[[(x & 255, gran * ((x >> 8) & 255), gran * ((x >> 16) & 255), (x >> 24) & 7, (x >> 27) & 15) for x in wave if (x >> 16) & 255]
                               for wave in wg] for wg in words]
        plan["merges"] = [[bool(wave[0] >> 31) for wave in wg] for wg in words]
        plan["raw_words"] = words
        return plan
    if plan["use_table"] == 3:
        # job lists (one-wave workgroups, the kernel's kList variant): "chain" = (b,h)s per period, "n_paired" = lists per
        # period (= grid_y), "simd_cap" = units of the longest list; per list its jobs as (units of the block = block index
        # + 1, (b,h) within the period); "descending_up_to" = blocks of at most that many units are taken from the diagonal
        # unit down.  Period P = 8 * z + x of the grid covers the (b,h)s [P * period_bh, (P + 1) *
        # period_bh) that exist; a list's entries whose (b,h) does not exist are dropped by the kernel.
        plan.update(family="lists", period_bh=plan["chain"], lists_per_period=plan["n_paired"], units_per_list=plan["simd_cap"],
                    jobs_per_list=plan["n_alone"], descending_up_to=plan["workspace_256"], workspace_256=0)
        words = [[buf[16 + l * 8 + i] & 0xffffffff for i in range(8)] for l in range(plan["lists_per_period"])]
        entries = [[(w >> (16 * h)) & 0xffff for w in ws for h in (0, 1)] for ws in words]
        plan["lists"] = [[(e & 255, e >> 8) for e in es if e & 255] for es in entries]
        plan["raw_words"] = words
        return plan
    plan["family"] = "units32"
    segs = plan["segs"]
    plan["workgroups"] = [[tuple(buf[16 + (j * segs + i) * 5: 16 + (j * segs + i) * 5 + 5]) for i in range(segs)
                           if buf[16 + (j * segs + i) * 5] != 0] for j in range(plan["wgs_per_bh"])]
    return plan


def plan_description(batch: int, seq_len: int, num_heads: int, workspace_bytes: int = 0) -> str:
    """One line on the plan the library uses for this shape on the current device (cmhsa_plan_describe)."""
    buf = ctypes.create_string_buffer(512)
    rc = load_library().cmhsa_plan_describe(batch, seq_len, num_heads, 1 if workspace_bytes > 0 else 0, buf, len(buf))
    if rc != CMHSA_OK:
        raise CmhsaError(rc, status_string(rc))
    return buf.value.decode()


class Workspace:
    """Device workspace of cmhsa_fp32_fwd_ws: CMHSA_WORKSPACE_BYTES of torch-allocated (hipMalloc'd) device memory whose
    counter block is zeroed once, on torch's current stream.  One per stream: calls that overlap in time on one GPU must
    not share it."""

    def __init__(self, device=None):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        lib = load_library()
        self.nbytes = int(lib.cmhsa_workspace_bytes())
        self.buffer = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
        if self.buffer.data_ptr() % 256:
            raise RuntimeError("allocator returned memory that is not 256-byte aligned")
        self.ptr = ctypes.c_void_p(self.buffer.data_ptr())
        self.device = device
        self.reset()

    def reset(self) -> None:
        """Zero the counter block again (only needed after a launch that did not run to completion)."""
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = load_library().cmhsa_workspace_init(self.ptr, self.nbytes, ctypes.c_void_p(stream))
        if rc != CMHSA_OK:
            raise CmhsaError(rc, status_string(rc))


def _check(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA tensor")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32")
    if t.dim() not in (2, 3):
        raise ValueError(f"{name} must be a 2D (N, d_model) or 3D (B, N, d_model) tensor")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}'s last-dim stride must be 1")


def attention_fwd_raw(q_ptr: int, k_ptr: int, v_ptr: int, o_ptr: int,
                      row_strides, batch_strides, batch: int, seq_len: int, num_heads: int, head_dim: int,
                      stream: int = 0) -> int:
    """Direct call of cmhsa_fp32_fwd with raw device addresses; returns the status code."""
    return load_library().cmhsa_fp32_fwd(
        q_ptr, k_ptr, v_ptr, o_ptr, *[int(s) for s in row_strides], *[int(s) for s in batch_strides],
        batch, seq_len, num_heads, head_dim, ctypes.c_void_p(stream))


def attention_fwd(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, num_heads: int,
                  out: torch.Tensor | None = None, workspace: "Workspace | None" = None) -> torch.Tensor:
    """Causal multi-head self-attention forward through the C ABI, on torch's current stream.

    Q, K, V: (N, d_model) or (B, N, d_model) fp32 device tensors, last-dim stride 1, arbitrary
    row stride (multiple of 4).  Returns a new contiguous tensor unless `out` is given.
    With `workspace` the call goes through cmhsa_fp32_fwd_ws (query blocks may be split across workgroups).
    """
    for t, n in ((Q, "Q"), (K, "K"), (V, "V")):
        _check(t, n)
    if K.shape != Q.shape or V.shape != Q.shape:
        raise ValueError("Q, K and V must have the same shape")
    d_model = Q.shape[-1]
    if num_heads <= 0 or d_model % num_heads != 0:
        raise ValueError("Feature dimension of Q must be evenly divisible by the number of heads")
    if out is None:
        out = torch.empty(Q.shape, dtype=torch.float32, device=Q.device)
    batched = Q.dim() == 3
    B = Q.shape[0] if batched else 1
    N = Q.shape[-2]
    bs = [t.stride(0) if batched else 0 for t in (Q, K, V, out)]
    rs = [t.stride(-2) for t in (Q, K, V, out)]
    with torch.cuda.device(Q.device):
        stream = torch.cuda.current_stream(Q.device).cuda_stream
        if workspace is not None:
            if workspace.device != Q.device:
                raise ValueError("workspace is on another device")
            rc = load_library().cmhsa_fp32_fwd_ws(
                Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), *[int(x) for x in rs], *[int(x) for x in bs],
                B, N, num_heads, d_model // num_heads, workspace.ptr, workspace.nbytes, ctypes.c_void_p(stream))
        else:
            rc = attention_fwd_raw(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), rs, bs,
                                   B, N, num_heads, d_model // num_heads, stream)
    if rc != CMHSA_OK:
        raise CmhsaError(rc, status_string(rc))
    return out


def attention_fwd_packed_qkv(QKV: torch.Tensor, num_heads: int, workspace: "Workspace | None" = None) -> torch.Tensor:
    """cmhsa_fp32_fwd_packed_qkv: QKV is the (N, 3*d_model) or (B, N, 3*d_model) output of the fused QKV linear layer
    (row = [Q | K | V], reference causal_self_attention.py:224-227); returns the (.., N, d_model) context."""
    _check(QKV, "QKV")
    if num_heads <= 0 or QKV.shape[-1] % (3 * num_heads) != 0:
        raise ValueError("last dimension of QKV must be 3 * d_model with d_model divisible by the number of heads")
    d_model = QKV.shape[-1] // 3
    out = torch.empty(QKV.shape[:-1] + (d_model,), dtype=torch.float32, device=QKV.device)
    batched = QKV.dim() == 3
    with torch.cuda.device(QKV.device):
        stream = torch.cuda.current_stream(QKV.device).cuda_stream
        rc = load_library().cmhsa_fp32_fwd_packed_qkv(
            QKV.data_ptr(), out.data_ptr(), QKV.stride(-2), out.stride(-2),
            QKV.stride(0) if batched else 0, out.stride(0) if batched else 0,
            QKV.shape[0] if batched else 1, QKV.shape[-2], num_heads, d_model // num_heads,
            workspace.ptr if workspace is not None else None, workspace.nbytes if workspace is not None else 0, ctypes.c_void_p(stream))
    if rc != CMHSA_OK:
        raise CmhsaError(rc, status_string(rc))
    return out


def decode_step(qkv_new: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, pos: torch.Tensor, num_heads: int,
                out: torch.Tensor | None = None) -> torch.Tensor:
    """cmhsa_fp32_decode_step (opt-in KV-cache mode): qkv_new (B, 3*d_model) of the new token, caches (B, capacity,
    d_model) updated in place at row `pos` (int32 device tensor, one element); returns the (B, d_model) context."""
    for t in (qkv_new, k_cache, v_cache):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1:
            raise ValueError("qkv_new, k_cache and v_cache must be float32 CUDA tensors with last-dim stride 1")
    if not pos.is_cuda or pos.dtype != torch.int32 or pos.numel() != 1:
        raise ValueError("pos must be an int32 CUDA tensor with one element")
    if qkv_new.dim() != 2 or k_cache.dim() != 3 or k_cache.shape != v_cache.shape or k_cache.stride() != v_cache.stride():
        raise ValueError("qkv_new must be (B, 3*d_model); k_cache and v_cache (B, capacity, d_model) with equal strides")
    B, d_model = qkv_new.shape[0], qkv_new.shape[1] // 3
    if num_heads <= 0 or qkv_new.shape[1] != 3 * d_model or d_model % num_heads or k_cache.shape[0] != B or k_cache.shape[2] != d_model:
        raise ValueError("shape mismatch between qkv_new, the caches and num_heads")
    if out is None:
        out = torch.empty((B, d_model), dtype=torch.float32, device=qkv_new.device)
    with torch.cuda.device(qkv_new.device):
        stream = torch.cuda.current_stream(qkv_new.device).cuda_stream
        rc = load_library().cmhsa_fp32_decode_step(
            qkv_new.data_ptr(), qkv_new.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), k_cache.stride(1), k_cache.stride(0),
            out.data_ptr(), out.stride(0), pos.data_ptr(), B, k_cache.shape[1], num_heads, d_model // num_heads,
            ctypes.c_void_p(stream))
    if rc != CMHSA_OK:
        raise CmhsaError(rc, status_string(rc))
    return out


def sustained_shader_clock_mhz(device=None, iters: int = 2000) -> float:
    """Shader clock (MHz) the chip sustains while every CU runs fp32 MFMA chains: cmhsa_probe_shader_clock."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.zeros(4, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        for _ in range(2):                                   # the first pass ramps the clocks, the second is read
            rc = load_library().cmhsa_probe_shader_clock(out.data_ptr(), iters, ctypes.c_void_p(stream))
            if rc != CMHSA_OK:
                raise CmhsaError(rc, status_string(rc))
        torch.cuda.synchronize(device)
    t0, t1, r0, r1 = (int(x) for x in out.tolist())
    return (t1 - t0) / max(r1 - r0, 1) * 100.0


def gelu_tanh_fwd(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """Fused tanh-GELU (reference gelu.py:8-12) through the C ABI of include/gpt2_gelu.h, on torch's current
    stream.  x: contiguous fp32 device tensor of any shape; `out` may be x itself (in place)."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise ValueError("x must be a float32 CUDA tensor")
    if not x.is_contiguous():
        raise ValueError("x must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != x.dtype or out.device != x.device or not out.is_contiguous():
        raise ValueError("out must match x (shape, dtype, device) and be contiguous")
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = load_library().gpt2_gelu_tanh_f32_fwd(x.data_ptr(), out.data_ptr(), x.numel(), ctypes.c_void_p(stream))
    if rc != CMHSA_OK:
        raise CmhsaError(rc, status_string(rc))
    return out
