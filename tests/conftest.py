import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import torch

    def unpack(o):
        if isinstance(o, torch.Tensor):
            return o.float() if o.dtype == torch.bfloat16 else o
        if isinstance(o, dict):
            return {k: unpack(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(unpack(v) for v in o)
        return o

    return unpack(torch.load(os.path.join(GOLDEN, name), weights_only=False))


@pytest.fixture(scope="session")
def golden_ops():
    return load_golden("ops_small.pt")
