"""Host-side mirror of the reference's GELU module (reference gelu.py:4-12) with the fused HIP kernel behind it
(SURVEY.md section 8(f) row 3).

    GELU()                      the reference's eight-op PyTorch expression (gelu.py:9-12), any device
    GELU(fused=True)            ONE launch of gpt2_gelu_tanh_f32_fwd (include/gpt2_gelu.h) via the torch op
                                `torch.ops.causal_multihead_self_attention.gelu_tanh`; device tensors only --
                                there is no CPU fallback, a CPU tensor raises.
Both compute y = 0.5*x*(1 + tanh(sqrt(2/pi)*(x + 0.44715*x^3))) with the reference's constant and operation order.
"""
from __future__ import annotations

import torch
from torch import nn


def gelu_reference_expression(x: torch.Tensor) -> torch.Tensor:
    """The reference's expression tree, op for op (gelu.py:9-12)."""
    return 0.5 * x * (1 + torch.tanh(torch.sqrt(torch.tensor(2.0 / torch.pi)) * (x + 0.44715 * torch.pow(x, 3))))


def _load_op():
    import causal_multihead_self_attention  # noqa: F401  (in-tree extension: registers the ops on import)

    return torch.ops.causal_multihead_self_attention.gelu_tanh


class GELU(nn.Module):
    def __init__(self, fused: bool = False):
        super().__init__()
        self.fused = fused
        self._op = _load_op() if fused else None     # fail at construction, loudly, if the extension is missing

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.fused:
            return gelu_reference_expression(x)
        if not x.is_cuda:
            raise RuntimeError("GELU(fused=True) runs the HIP kernel only: the input must be a device tensor")
        return self._op(x)
