"""pytest configuration: the `gpu` marker and import paths.

`python -m pytest tests -m "not gpu"` runs on a CPU-only box: oracle vs golden vectors, host
logic, the C-ABI library's exports, gloo multi-process paths.  `-m gpu` tests are the parity
tests proper: they run the HIP kernels on a real MI355X through the C ABI and the torch op.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "llm-inference_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible (the HIP path has no CPU fallback)")
    return torch.device("cuda:0")
