import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def _limit_cpu_threads():
    # the CPU oracle (oneDNN) is several times slower with 128+ threads than with 16 on the GPU box's 256-thread host
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass


_limit_cpu_threads()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("filterwarnings", "ignore:The argument 'device' of Tensor:DeprecationWarning")     # torch's own DataLoader pin_memory path


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def rel_linf(a, b):
    """max|a-b| / max|b|  -- the max-norm-relative metric of SURVEY.md §8c."""
    import torch
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
