import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "slow: multi-second CPU oracle runs")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_l2(a, b):
    import torch
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def hin(name, shape, scale=1.0):
    """Hash-generated input; identical to tools/gen_golden.py:hin."""
    import torch
    from sd_amd import synth
    return torch.from_numpy(synth.hash_uniform("input:" + name, int(np.prod(shape))).reshape(shape)) * scale
