"""Generator of tests/golden/gelu_*.npz.  AUTHORING CONTAINER ONLY (imports the reference's gelu.py from
/root/reference: torch-only, runs on CPU).  Writes the reference module's OUTPUTS for the inputs that
gelu_cases.py regenerates from seeds; nothing from the reference is copied.  Also checks the repo's own oracle
(oracle/gelu.py) against the reference on every case.

Usage (from the repo root):  python tests/golden/make_golden_gelu.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import gelu_cases  # noqa: E402
from oracle import gelu as oracle  # noqa: E402
from gelu import GELU  # noqa: E402  (the reference module)


def main():
    ref = GELU()
    for name in gelu_cases.GELU_CASES:
        x = gelu_cases.make_input(name)
        with torch.no_grad():
            expected = ref(torch.from_numpy(x)).numpy()
        np.savez_compressed(gelu_cases.golden_path(name), expected=expected)
        mine = oracle.gelu_tanh_torch(torch.from_numpy(x)).numpy()
        same = np.array_equal(mine, expected, equal_nan=True)
        f64 = oracle.gelu_tanh_f64(x)
        fin = np.isfinite(expected) & np.isfinite(f64)
        print(f"{name:16s} n={x.size:7d} oracle-torch bit-equal to reference: {same}; "
              f"max |ref - f64| = {np.abs(expected[fin] - f64[fin]).max():.3e}")
        assert same


if __name__ == "__main__":
    main()
