"""ORACLE (test infrastructure, not product code): residual add + LayerNorm, restating torch.nn.LayerNorm as the reference
uses it (reference transformer_block.py:11, :22, :26-39; gpt2.py:37): biased variance over the last dimension, eps inside
the square root, elementwise affine.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned in tests/test_layernorm.py against torch.nn.functional.layer_norm on the CPU (float64 and float32) -- the very
function the reference's modules call."""
from __future__ import annotations

import numpy as np


def add_layer_norm_f64(x, delta, weight, bias, eps: float = 1e-5):
    """Returns (x + delta, LayerNorm(x + delta)) with the add in float32 (as the reference's `x = x + ...` on fp32
    tensors) and the normalisation in float64.  delta None: (x, LayerNorm(x))."""
    s = np.asarray(x, dtype=np.float32)
    if delta is not None:
        s = (s + np.asarray(delta, dtype=np.float32)).astype(np.float32)
    s64 = s.astype(np.float64)
    mean = s64.mean(axis=-1, keepdims=True)
    var = ((s64 - mean) ** 2).mean(axis=-1, keepdims=True)
    y = (s64 - mean) / np.sqrt(var + eps) * np.asarray(weight, dtype=np.float64) + np.asarray(bias, dtype=np.float64)
    return s, y
