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


@pytest.fixture(scope="session")
def models():
    """The four product modules with hash weights on the HIP device (shared by every GPU test module: building and
    hashing ~1.07 B parameters takes tens of seconds)."""
    import torch
    from sd_amd import clip, decoder, diffusion, encoder, synth
    torch.set_grad_enabled(False)
    mk = lambda cls, prefix: synth.load_hashed(cls(), prefix).to("cuda")
    return {
        "clip": mk(clip.CLIP, "clip."),
        "encoder": mk(encoder.VAE_Encoder, "encoder."),
        "decoder": mk(decoder.VAE_Decoder, "decoder."),
        "diffusion": mk(diffusion.Diffusion, "diffusion."),
    }


_CPU_SDS = {}


def cpu_state_dicts(models):
    """CPU copies of the product modules' state_dicts (what the oracle consumes), made once."""
    if not _CPU_SDS:
        _CPU_SDS.update({k: {n: t.cpu() for n, t in m.state_dict().items()} for k, m in models.items()})
    return _CPU_SDS
