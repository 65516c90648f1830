"""``preload_models_from_standard_weights`` with the reference's signature (loader.py:10-34):
one CompVis-layout checkpoint -> ``{"clip", "encoder", "decoder", "diffusion"}`` modules of this
package, resident on the HIP device, weights loaded with ``strict=True``.

Parameters keep the checkpoint values in float32 (an fp16 checkpoint is widened on load); the
kernels' private layouts (bf16 copies, fused K|V, phase-summed upsample taps ...) are packed
lazily on first use and cached (engine.packed), so loading itself launches nothing.
"""
import torch

from . import model_converter
from ._lib import SdamdError
from .clip import CLIP
from .decoder import VAE_Decoder
from .diffusion import Diffusion
from .encoder import VAE_Encoder


def _materialise(cls, state_dict, device, **kw):
    """Build ``cls`` without initialising its parameters (meta device), then adopt the checkpoint
    tensors: no throw-away random init of ~1 B parameters, one host->device copy per tensor."""
    with torch.device("meta"):
        module = cls(**kw)
    module.to_empty(device=device)
    module.load_state_dict(state_dict, strict=True)
    for p in module.parameters():
        p.requires_grad_(False)
    return module.eval()


def preload_models_from_standard_weights(ckpt_path, vae_checkpoint_path=None, attention_type="xformers",
                                         device="cuda", trusted=False):
    device = torch.device(device)
    if device.type != "cuda":
        raise SdamdError("sd_amd.loader places the models on the HIP device only (device='cuda'); no CPU fallback")
    if vae_checkpoint_path:
        # loader.py:13-14 calls vae.load_vae_models, which the reference's vae.py does not define
        # (AttributeError there); the side-project VAE (vae.py:28-110) is outside this path.
        raise NotImplementedError("vae_checkpoint_path: the reference's vae.load_vae_models does not exist; "
                                  "the VAE comes from the main checkpoint")
    state_dict = model_converter.load_from_standard_weights(ckpt_path, device, trusted=trusted)
    return {
        "clip": _materialise(CLIP, state_dict["clip"], device, attention_type=attention_type),
        "encoder": _materialise(VAE_Encoder, state_dict["encoder"], device),
        "decoder": _materialise(VAE_Decoder, state_dict["decoder"], device),
        "diffusion": _materialise(Diffusion, state_dict["diffusion"], device, attention_type=attention_type),
    }
