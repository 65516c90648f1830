from .ddim import DDIMSampler
from .ddpm import DDPMSampler

__all__ = ["DDPMSampler", "DDIMSampler"]
