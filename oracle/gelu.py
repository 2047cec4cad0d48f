"""CPU restatement of the reference's GELU (reference gelu.py:4-12).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path.

    y = 0.5 * x * (1 + tanh( sqrt(2/pi) * (x + 0.44715 * x^3) ))

The constant 0.44715 is the reference's (gelu.py:11), not the textbook 0.044715.  Pinned against the reference
module run in the authoring container: tests/golden/make_golden_gelu.py, tests/golden/gelu_*.npz (bit-equal).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def gelu_tanh_torch(x: torch.Tensor) -> torch.Tensor:
    """fp32 torch evaluation in the reference's operation order (gelu.py:9-12): every product and sum is a
    separate rounded fp32 operation, the 0-dim sqrt(2/pi) tensor acts as an fp32 scalar."""
    c = torch.sqrt(torch.tensor(2.0 / torch.pi))
    inner = x + 0.44715 * torch.pow(x, 3)
    return 0.5 * x * (1 + torch.tanh(c * inner))


def gelu_tanh_f64(x: np.ndarray) -> np.ndarray:
    """The same formula in float64 (what the fp32 chain approximates)."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(over="ignore", invalid="ignore"):
        return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.44715 * x ** 3)))
