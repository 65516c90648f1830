from .ddim import DDIMSampler
from .ddpm import DDPMSampler
from .euler import EulerSampler

__all__ = ["DDPMSampler", "DDIMSampler", "EulerSampler"]
