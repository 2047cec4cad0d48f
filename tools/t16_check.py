"""Quick GPU check of the 16-row-block family (cmhsa_fwd16.hip), forced on through the plan-forcing test build:
    CMHSA_T16=1 python tools/t16_check.py
Compares with the f64 oracle on ragged / exact lengths, several head counts and batches, rand / randn / spiky inputs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "llm-inference_amd"))
import numpy as np
import torch

import cmhsa

cmhsa.LIB_PATH = os.path.join(ROOT, "tests", "helpers", "libcmhsa_hip_planenv.so")
from oracle import naive_attention as oracle

dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
worst = 0.0
n16_seen = 0
for (B, H, T, dist) in [(1, 12, 1, "randn"), (1, 12, 7, "randn"), (1, 12, 16, "randn"), (1, 12, 17, "rand"), (1, 12, 31, "randn"), (1, 12, 32, "randn"),
                        (1, 12, 33, "randn"), (1, 12, 48, "randn"), (1, 12, 100, "spiky"), (1, 12, 128, "rand"), (1, 12, 200, "randn"), (1, 12, 256, "randn"),
                        (1, 12, 300, "spiky"), (1, 12, 511, "randn"), (1, 12, 512, "rand"), (1, 12, 768, "randn"), (1, 12, 1000, "spiky"), (1, 12, 1024, "rand"),
                        (1, 25, 512, "randn"), (1, 16, 777, "randn"), (2, 12, 300, "randn"), (3, 5, 129, "randn"), (1, 1, 2048, "randn"), (1, 3, 4080, "randn"),
                        (4, 16, 1024, "randn"), (8, 12, 64, "randn"), (1, 20, 1024, "randn"), (16, 16, 100, "spiky")]:
    d = 64 * H
    buf = rng.standard_normal((B, T, 3 * d), dtype=np.float32) if dist != "rand" else rng.random((B, T, 3 * d), dtype=np.float32)
    if dist == "spiky":                       # forces the running-max jump late in a row (online-softmax rescale)
        buf[:, T // 2:, d:2 * d] *= 6.0
        buf[:, -1, d:2 * d] *= 3.0
    x = torch.from_numpy(buf).to(dev)
    q, k, v = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    desc = cmhsa.plan_description(B, T, H, 0)
    out = torch.full((B, T, d), float("nan"), device=dev)
    cmhsa.attention_fwd(q, k, v, H, out=out)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.isfinite(o).all(), (B, H, T, "unwritten or non-finite output")
    err = 0.0
    for b in sorted({0, B - 1}):
        for h in sorted({0, H // 2, H - 1}):
            sl = slice(64 * h, 64 * h + 64)
            want = oracle.naive_attention_f64(buf[b, :, sl], buf[b, :, d + 64 * h:d + 64 * h + 64], buf[b, :, 2 * d + 64 * h:2 * d + 64 * h + 64], 1)
            err = max(err, float(np.abs(o[b, :, sl] - want).max()))
    is16 = desc.startswith("16-key tiles")
    n16_seen += is16
    print(f"B={B} H={H} T={T} {dist}: max err {err:.2e}  [{desc.split(':')[0]}]", flush=True)
    worst = max(worst, err)
    assert err <= 1e-4, (B, H, T, err)
    again = cmhsa.attention_fwd(q, k, v, H)
    assert torch.equal(again, out), "not reproducible"
print("worst", worst, "cases on the 16-row family:", n16_seen)
