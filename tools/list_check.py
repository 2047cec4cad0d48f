"""GPU check of the job-list family (one-wave workgroups that run lists of whole query blocks of several (b,h)s), forced on
through the plan-forcing test build:   CMHSA_LIST=1 python tools/list_check.py
Every output element against torch's fp32 math attention on the GPU (all (b,h)s: a wrong (b,h) base or a dropped job shows),
sampled (b,h)s against the f64 oracle, QKV-sliced inputs, ragged lengths, (b,h) counts that leave a short last period."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "llm-inference_amd"))
import numpy as np
import torch
import torch.nn.functional as F

import cmhsa

cmhsa.LIB_PATH = os.path.join(ROOT, "tests", "helpers", "libcmhsa_hip_planenv.so")
from oracle import naive_attention as oracle

dev = torch.device("cuda:0")
worst = 0.0
seen = 0
g = torch.Generator(device=dev).manual_seed(5)
for (B, H, T, dist) in [(64, 25, 512, "randn"), (32, 25, 512, "rand"), (16, 25, 512, "randn"), (64, 25, 256, "spiky"), (64, 25, 500, "randn"),
                        (37, 25, 511, "randn"), (64, 12, 1024, "rand"), (100, 7, 300, "spiky"), (64, 25, 128, "randn"), (64, 25, 65, "randn"),
                        (333, 3, 200, "randn"), (41, 16, 777, "randn"), (64, 25, 384, "rand"), (1000, 1, 480, "randn"), (16, 16, 2048, "randn")]:
    d = 64 * H
    x = (torch.randn if dist != "rand" else torch.rand)((B, T, 3 * d), device=dev, generator=g)
    if dist == "spiky":
        x[:, T // 2:, d:2 * d] *= 6.0
        x[:, -1, d:2 * d] *= 3.0
    q, k, v = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    desc = cmhsa.plan_description(B, T, H, 0)
    out = torch.full((B, T, d), float("nan"), device=dev)
    cmhsa.attention_fwd(q, k, v, H, out=out)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all(), (B, H, T, "unwritten or non-finite output")
    split = lambda t: t.reshape(B, T, H, 64).transpose(1, 2)
    with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.MATH):
        ref = F.scaled_dot_product_attention(split(q), split(k), split(v), is_causal=True).transpose(1, 2).reshape(B, T, d)
    err_t = float((out - ref).abs().max())
    buf = x.cpu().numpy()
    o = out.cpu().numpy()
    err = 0.0
    for b in sorted({0, B // 3, B - 1}):
        for h in sorted({0, H // 2, H - 1}):
            sl = slice(64 * h, 64 * h + 64)
            want = oracle.naive_attention_f64(buf[b, :, sl], buf[b, :, d + 64 * h:d + 64 * h + 64], buf[b, :, 2 * d + 64 * h:2 * d + 64 * h + 64], 1)
            err = max(err, float(np.abs(o[b, :, sl] - want).max()))
    is_list = desc.startswith("job lists")
    seen += is_list
    print(f"B={B} H={H} T={T} {dist}: vs torch {err_t:.2e}  vs f64 oracle {err:.2e}  [{desc.split(', workspace')[0]}]", flush=True)
    worst = max(worst, err)
    assert err <= 1e-4 and err_t <= 2e-4, (B, H, T, err, err_t)
    again = cmhsa.attention_fwd(q, k, v, H)
    assert torch.equal(again, out), "not reproducible"
    del x, out, ref, again
print("worst", worst, "cases on job lists:", seen)
