"""Golden-vector case table shared by make_golden.py (the generator, authoring container only)
and the tests (which only read the committed .npz files).

Inputs are NOT stored: they are regenerated from `numpy.random.default_rng(seed)` (PCG64, whose
stream numpy guarantees stable across versions), so a fixture holds only expected outputs.
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# name -> dict(seed, T, d_model, H, dist, layout, rows)
#   dist   : "rand"  = U[0,1) like the reference test (test_torch_extension.py:16-18)
#            "randn" = N(0,1), numerically harsher (peaked softmax)
#   layout : "contig" = three independent (T, d_model) arrays
#            "qkv"    = column slices of one (T, 3*d_model) buffer, row stride 3*d_model, exactly
#                       what CausalSelfAttention.forward hands over (causal_self_attention.py:224-227)
#   rows   : None = all rows stored; else an int stride: rows[::stride] plus the last row
ATTENTION_CASES = {
    "t1_d768_rand":        dict(seed=101, T=1,    d_model=768,  H=12, dist="rand",  layout="contig", rows=None),
    "t7_d768_rand":        dict(seed=102, T=7,    d_model=768,  H=12, dist="rand",  layout="contig", rows=None),
    "t33_d768_rand":       dict(seed=103, T=33,   d_model=768,  H=12, dist="rand",  layout="contig", rows=None),
    "t64_d768_rand":       dict(seed=104, T=64,   d_model=768,  H=12, dist="rand",  layout="contig", rows=None),
    "t65_d768_randn":      dict(seed=105, T=65,   d_model=768,  H=12, dist="randn", layout="contig", rows=None),
    "t129_d768_randn":     dict(seed=106, T=129,  d_model=768,  H=12, dist="randn", layout="contig", rows=None),
    "t199_d768_rand":      dict(seed=107, T=199,  d_model=768,  H=12, dist="rand",  layout="contig", rows=None),
    "t100_d768_qkv_randn": dict(seed=108, T=100,  d_model=768,  H=12, dist="randn", layout="qkv",    rows=None),
    "t48_d1024_randn":     dict(seed=109, T=48,   d_model=1024, H=16, dist="randn", layout="qkv",    rows=None),
    "t48_d1280_randn":     dict(seed=110, T=48,   d_model=1280, H=20, dist="randn", layout="contig", rows=None),
    "t48_d1600_randn":     dict(seed=111, T=48,   d_model=1600, H=25, dist="randn", layout="qkv",    rows=None),
    "t1024_d768_rand":     dict(seed=112, T=1024, d_model=768,  H=12, dist="rand",  layout="contig", rows=41),
    "t1024_d768_randn":    dict(seed=113, T=1024, d_model=768,  H=12, dist="randn", layout="qkv",    rows=41),
}

# CausalSelfAttention module case (fused QKV linear -> slices -> attention -> out-proj,
# reference causal_self_attention.py:222-236), weights drawn from the same numpy stream.
MODULE_CASE = dict(seed=201, T=40, d_model=768, H=12)


def make_inputs(case: dict):
    """Return (Q, K, V) float32 numpy arrays of shape (T, d_model); for layout "qkv" they are
    non-contiguous views into one (T, 3*d_model) array."""
    rng = np.random.default_rng(case["seed"])
    T, d = case["T"], case["d_model"]
    draw = (lambda shape: rng.random(shape, dtype=np.float32)) if case["dist"] == "rand" \
        else (lambda shape: rng.standard_normal(shape, dtype=np.float32))
    if case["layout"] == "qkv":
        buf = draw((T, 3 * d))
        return buf[:, 0:d], buf[:, d:2 * d], buf[:, 2 * d:3 * d]
    return draw((T, d)), draw((T, d)), draw((T, d))


def stored_rows(case: dict) -> np.ndarray:
    T = case["T"]
    if case["rows"] is None:
        return np.arange(T)
    idx = list(range(0, T, case["rows"]))
    if idx[-1] != T - 1:
        idx.append(T - 1)
    return np.asarray(idx)


def make_module_params(case: dict):
    """x (T,d) and the CausalSelfAttention parameters, all from one numpy stream."""
    rng = np.random.default_rng(case["seed"])
    T, d = case["T"], case["d_model"]
    n = lambda *shape, s=1.0: (rng.standard_normal(shape, dtype=np.float32) * np.float32(s))
    return dict(
        x=n(T, d),
        qkv_weight=n(3 * d, d, s=d ** -0.5),
        qkv_bias=n(3 * d, s=0.1),
        out_weight=n(d, d, s=d ** -0.5),
        out_bias=n(d, s=0.1),
    )


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name + ".npz")
