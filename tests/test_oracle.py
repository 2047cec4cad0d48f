"""CPU tests: the oracle (oracle/) against the golden vectors produced by the reference module.

The golden outputs in tests/golden/*.npz come from the reference's own
scaled_dot_product_attention(NAIVE_PYTORCH, ...) (reference causal_self_attention.py:95-126,
:173-194), run in the authoring container by tests/golden/make_golden.py.
"""
import numpy as np
import pytest
import torch

import cases
from oracle import naive_attention as oracle

SMALL = [n for n, c in cases.ATTENTION_CASES.items() if c["T"] <= 199]
ALL = list(cases.ATTENTION_CASES)


def _golden(name):
    z = np.load(cases.golden_path(name))
    return z["rows"], z["expected"]


@pytest.mark.parametrize("name", ALL)
def test_oracle_torch_matches_reference_bitwise(name):
    case = cases.ATTENTION_CASES[name]
    Q, K, V = cases.make_inputs(case)
    rows, expected = _golden(name)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).unsqueeze(0)
    got = oracle.naive_attention_torch(t(Q), t(K), t(V), case["H"]).squeeze(0).numpy()
    # same op sequence as the reference on the same torch build: equal to the last bit here; a
    # different torch/BLAS build may differ in the last ulps, hence the tiny bound instead of ==
    assert np.abs(got[rows] - expected).max() <= 2e-6


@pytest.mark.parametrize("name", SMALL)
def test_oracle_f64_close_to_reference(name):
    case = cases.ATTENTION_CASES[name]
    Q, K, V = cases.make_inputs(case)
    rows, expected = _golden(name)
    got = oracle.naive_attention_f64(Q, K, V, case["H"])
    assert np.abs(got[rows] - expected).max() <= 5e-6


@pytest.mark.parametrize("name", SMALL)
def test_oracle_c_close_to_reference(name):
    case = cases.ATTENTION_CASES[name]
    Q, K, V = cases.make_inputs(case)
    rows, expected = _golden(name)
    got = oracle.naive_attention_c(Q, K, V, case["H"], threads=4)
    assert np.abs(got[rows] - expected).max() <= 5e-6


def test_oracle_c_strided_and_batched():
    rng = np.random.default_rng(7)
    B, T, H = 2, 37, 3
    d = 64 * H
    buf = rng.standard_normal((B, T, 3 * d), dtype=np.float32)
    q, k, v = buf[..., :d], buf[..., d:2 * d], buf[..., 2 * d:]
    ref = oracle.naive_attention_f64(q, k, v, H)
    got = oracle.naive_attention_c(q, k, v, H, threads=2)
    assert got.shape == (B, T, d)
    assert np.abs(got - ref).max() <= 5e-6


def test_first_row_is_first_value_row():
    # N = 1 -> O = V[0]; and row 0 of any length attends to key 0 only (SURVEY 8(a) edge cases)
    case = cases.ATTENTION_CASES["t7_d768_rand"]
    Q, K, V = cases.make_inputs(case)
    out = oracle.naive_attention_f64(Q, K, V, case["H"])
    assert np.allclose(out[0], V[0], atol=1e-7)
    out1 = oracle.naive_attention_f64(Q[:1], K[:1], V[:1], case["H"])
    assert np.allclose(out1[0], V[0], atol=1e-7)


def test_indivisible_heads_rejected():
    x = np.zeros((4, 100), dtype=np.float32)
    with pytest.raises(ValueError):
        oracle.naive_attention_f64(x, x, x, 3)
