from .ddpm import DDPMSampler

__all__ = ["DDPMSampler"]
