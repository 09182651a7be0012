import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_runtest_setup(item):
    """-m gpu tests must not silently pass or skip on a box without a GPU: they fail loudly instead (the product has no
    CPU path, so a GPU test that ran on the CPU would have exercised nothing)."""
    if item.get_closest_marker("gpu") is not None:
        import torch
        if not torch.cuda.is_available():
            pytest.fail("gpu-marked test selected on a machine without a GPU (run `-m \"not gpu\"` here)", pytrace=False)
