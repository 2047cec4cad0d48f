"""Inputs of the GELU golden vectors (tests/golden/gelu_*.npz hold the reference module's outputs for them).
Inputs are regenerated from seeds; only expected outputs are stored."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

GELU_CASES = {
    # name: (seed, shape, kind)
    "randn_mlp":   (11, (37, 3072), "randn"),        # a GPT2-small MLP activation, ragged row count
    "wide_range":  (12, (4099,), "wide"),            # |x| up to 40: both saturated tails, odd length (scalar tail)
    "specials":    (13, (64,), "specials"),          # zeros, denormals, huge values, infinities, NaN
}


def golden_path(name: str) -> str:
    return os.path.join(HERE, f"gelu_{name}.npz")


def make_input(name: str) -> np.ndarray:
    seed, shape, kind = GELU_CASES[name]
    rng = np.random.default_rng(seed)
    if kind == "randn":
        return rng.standard_normal(shape).astype(np.float32)
    if kind == "wide":
        return (rng.uniform(-40.0, 40.0, shape)).astype(np.float32)
    base = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, 1e-20, 1.0, -1.0, 0.5, -0.5, 3.0, -3.0, 10.0, -10.0,
                     1e10, -1e10, 1e13, -1e13, 3e38, -3e38, np.inf, -np.inf, np.nan], dtype=np.float32)
    out = np.zeros(shape, dtype=np.float32)
    out[:base.size] = base
    out[base.size:] = rng.standard_normal(shape[0] - base.size).astype(np.float32) * 5
    return out
