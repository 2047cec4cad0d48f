"""Generator of the golden vectors in this directory.  AUTHORING CONTAINER ONLY.

It imports the reference's own Python oracle, `causal_self_attention.py`, from /root/reference
(torch-only imports; the CUDA extension is imported lazily and never touched here), runs
`scaled_dot_product_attention(NAIVE_PYTORCH, ...)` (reference causal_self_attention.py:95-126,
:173-194) and `CausalSelfAttention.forward` (:222-236) on CPU on the inputs of cases.py, and
writes only the expected OUTPUTS as .npz.  Nothing from the reference is copied: the fixtures
are data.  /root/reference does not exist on the GPU box; tests only read the .npz files.

Also cross-checks the repo's own oracle (oracle/naive_attention.py) against the reference on
every case and prints the max deviation -- this is the "oracle pinned by the reference run
here" check.

Usage (from the repo root):  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import cases  # noqa: E402
from oracle import naive_attention as oracle  # noqa: E402

import causal_self_attention as ref  # noqa: E402  (the reference module)


def ref_attention(Q, K, V, H):
    T = Q.shape[0]
    mask = torch.triu(torch.ones(T, T), diagonal=1)
    out = ref.scaled_dot_product_attention(
        ref.ScaledDotProductAttentionBackend.NAIVE_PYTORCH,
        queries=torch.from_numpy(Q).unsqueeze(0), keys=torch.from_numpy(K).unsqueeze(0),
        values=torch.from_numpy(V).unsqueeze(0), num_heads=H, causal_mask=mask)
    return out.squeeze(0).numpy()


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    worst_torch = worst_f64 = worst_c = 0.0
    for name, case in cases.ATTENTION_CASES.items():
        Q, K, V = cases.make_inputs(case)
        expected = ref_attention(Q, K, V, case["H"])
        rows = cases.stored_rows(case)
        np.savez_compressed(cases.golden_path(name), rows=rows.astype(np.int32),
                            expected=expected[rows].astype(np.float32))
        mine = oracle.naive_attention_torch(torch.from_numpy(np.ascontiguousarray(Q)).unsqueeze(0),
                                            torch.from_numpy(np.ascontiguousarray(K)).unsqueeze(0),
                                            torch.from_numpy(np.ascontiguousarray(V)).unsqueeze(0),
                                            case["H"]).squeeze(0).numpy()
        f64 = oracle.naive_attention_f64(Q, K, V, case["H"])
        e_t = float(np.abs(mine - expected).max())
        e_f = float(np.abs(f64 - expected).max())
        try:
            e_c = float(np.abs(oracle.naive_attention_c(Q, K, V, case["H"], threads=8) - expected).max())
        except FileNotFoundError:
            e_c = float("nan")
        worst_torch, worst_f64, worst_c = max(worst_torch, e_t), max(worst_f64, e_f), max(worst_c, e_c)
        print(f"{name:24s} rows={len(rows):4d} |oracle_torch-ref|={e_t:.2e} |oracle_f64-ref|={e_f:.2e} "
              f"|oracle_c-ref|={e_c:.2e}")

    # CausalSelfAttention module (fused QKV linear -> strided slices -> attention -> out-proj)
    mc = cases.MODULE_CASE
    p = cases.make_module_params(mc)
    mod = ref.CausalSelfAttention(d_in=mc["d_model"], d_out=mc["d_model"], max_context_length=1024,
                                  p_dropout=0.1, num_heads=mc["H"], qkv_bias=True,
                                  scaled_dot_product_attention_backend=ref.ScaledDotProductAttentionBackend.NAIVE_PYTORCH)
    mod.eval()
    with torch.no_grad():
        mod.qkv_layer.weight.copy_(torch.from_numpy(p["qkv_weight"]))
        mod.qkv_layer.bias.copy_(torch.from_numpy(p["qkv_bias"]))
        mod.out_proj_layer.weight.copy_(torch.from_numpy(p["out_weight"]))
        mod.out_proj_layer.bias.copy_(torch.from_numpy(p["out_bias"]))
        y = mod(torch.from_numpy(p["x"]).unsqueeze(0)).squeeze(0).numpy()
        qkv = mod.qkv_layer(torch.from_numpy(p["x"]).unsqueeze(0))
        q_stride = qkv[:, :, 0:mc["d_model"]].squeeze(0).stride()
    np.savez_compressed(cases.golden_path("module_t40_d768"), expected=y.astype(np.float32),
                        q_stride=np.asarray(q_stride, dtype=np.int64))
    print(f"module_t40_d768: out {y.shape}, q slice stride {q_stride}")
    print(f"worst |oracle_torch-ref|={worst_torch:.2e}  |oracle_f64-ref|={worst_f64:.2e}  |oracle_c-ref|={worst_c:.2e}")
    print(f"torch {torch.__version__}, numpy {np.__version__}")


if __name__ == "__main__":
    main()
