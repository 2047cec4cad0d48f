"""ORACLE (test infrastructure, not product code): a linear layer as the reference uses torch.nn.Linear around the
attention op (reference causal_self_attention.py:213-214, :224, :236; transformer_block.py:17-20, :35-37), optionally
followed by the reference's tanh-GELU (gelu.py:8-12, restated in oracle/gelu.py):

    y[m, n] = act( sum_k x[m, k] * weight[n, k] + bias[n] )

evaluated in float64.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
Pinned in tests/test_linear_rows.py against torch.nn.functional.linear on the CPU -- the very function the reference's
modules call."""
from __future__ import annotations

import numpy as np

from oracle.gelu import gelu_tanh_f64


def linear_f64(x, weight, bias=None, gelu: bool = False):
    y = np.asarray(x, dtype=np.float64) @ np.asarray(weight, dtype=np.float64).T
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)
    return gelu_tanh_f64(y) if gelu else y


def sum_scale(x, weight, bias=None):
    """sum_k |x[m,k] * weight[n,k]| + |bias[n]|: what one fp32 rounding error of the sum is relative to."""
    s = np.abs(np.asarray(x, dtype=np.float64)) @ np.abs(np.asarray(weight, dtype=np.float64)).T
    return s + (np.abs(np.asarray(bias, dtype=np.float64)) if bias is not None else 0.0)
