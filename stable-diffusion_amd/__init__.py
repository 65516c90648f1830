"""MI355X-native SD-v1.5 denoising hot path behind the reference's module interfaces.

Sub-modules mirror the reference's ``src/`` layout (``pipeline``, ``diffusion``, ``clip``,
``encoder``, ``decoder``, ``attention``, ``flash_attention``, ``sampler.ddpm`` / ``.ddim`` / ``.euler``, ``model_converter``,
``loader``).  All compute
goes through the C-ABI library ``csrc/libsdamd.so`` (hand-written gfx950 HIP); there is no
CPU or PyTorch-eager fallback: without the library every op raises.
"""
__all__ = ["synth"]


def install_as_reference_modules():
    """Register this package's modules under the reference's top-level names
    (``import pipeline``, ``import diffusion`` ...) so reference callers run unchanged."""
    import importlib
    import sys
    for name in ("attention", "flash_attention", "clip", "diffusion", "encoder", "decoder",
                 "pipeline", "sampler", "sampler.ddpm", "sampler.ddim", "sampler.euler", "model_converter", "loader"):
        sys.modules[name] = importlib.import_module("sd_amd." + name)
