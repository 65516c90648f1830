"""Deterministic DDIM sampler (eta = 0) behind the reference's ``DDIMSampler`` interface
(sampler/ddim.py:8-93).  The reference ships the class but its pipeline never imports it
(pipeline.py:4-5), so ``generate(sampler="ddim")`` raises there; here it is wired in
(SURVEY.md §8f rank 3).

Differences from DDPM that the reference's class defines and this one keeps: a plain linear beta
schedule 1e-4 .. 0.02 (ddim.py:15-20, NOT the scaled-linear one SD v1.5 was trained with), a default
of 50 steps set in the constructor (:29-30), ``set_strength`` keeping ``timesteps >= start_step``
(:42-45), previous timestep clamped at 0 (:38-40), no per-step noise.

The step runs in the same fused HIP kernel as DDPM (``sdamd_cfg_ddpm_step``): its coefficient row
carries ``cx_on_eps = 1`` so the second term weighs the predicted noise instead of the sample.
With no noise to pre-draw, the captured step graph reads only the latents and the coefficient table.
"""
import numpy as np
import torch

from .. import ops


def unnormalize_to_zero_to_one(t):
    """ddim.py:4-6"""
    return (t + 1) * 0.5


class DDIMSampler:
    stochastic = False      # no generator draws inside the step loop

    def __init__(self, generator: torch.Generator, training_steps=1000):
        scale = 1000 / training_steps
        self.betas = torch.linspace(scale * 0.0001, scale * 0.02, training_steps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alpha_cumprod = torch.cumprod(self.alphas, 0)
        self.one = torch.tensor(1.0)
        self.generator = generator
        self.training_steps = training_steps
        self.timesteps = torch.from_numpy(np.arange(0, training_steps)[::-1].copy())
        self.steps = 50
        self.set_steps(self.steps)

    # ---- schedule (host, bit-exact with the reference) -----------------------------------
    def set_steps(self, steps):
        self.steps = steps
        ratio = self.training_steps // steps
        self.timesteps = torch.from_numpy((np.arange(0, steps) * ratio).round()[::-1].copy().astype(np.int64))

    def _get_previous_timestep(self, timestep: int) -> int:
        return max(int(timestep) - self.training_steps // self.steps, 0)

    def set_strength(self, strength=0.7):
        start_step = self.training_steps - int(self.training_steps * strength)
        self.timesteps = self.timesteps[self.timesteps >= start_step]
        self.start_step = start_step

    def set_strength_steps(self, strength=0.7):
        """Intended img2img semantics (extension, same as DDPMSampler.set_strength_steps): run the LAST
        ``int(steps*strength)`` inference steps.  The reference's filter above keeps the high-noise end
        instead (strength 0.8 -> t = 980..200: starts from pure-noise levels and stops before t = 0)."""
        start_step = self.steps - int(self.steps * strength)
        self.timesteps = self.timesteps[start_step:]
        self.start_step = start_step

    def _get_variance(self, timestep: int) -> torch.Tensor:
        return torch.tensor(0.0)

    def step_coefficients(self, timestep) -> torch.Tensor:
        """f32[8] = {sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev), 0, 0, 1, 0}
        (ddim.py:55-70) in IEEE float32 scalar arithmetic."""
        f = np.float32
        t = int(timestep)
        a_t = f(self.alpha_cumprod[t].item())
        a_prev = f(self.alpha_cumprod[self._get_previous_timestep(t)].item())
        vals = [np.sqrt(a_t), np.sqrt(f(1) - a_t), np.sqrt(a_prev), np.sqrt(f(1) - a_prev), f(0), f(0), f(1), f(0)]
        return torch.from_numpy(np.asarray(vals, dtype=np.float32))

    def coefficient_table(self, timesteps=None) -> torch.Tensor:
        ts = self.timesteps if timesteps is None else timesteps
        return torch.stack([self.step_coefficients(t) for t in ts])

    def draw_noise(self, shape, dtype=torch.float32):
        return torch.randn(shape, generator=self.generator, device=self.generator.device, dtype=dtype)

    # ---- tensor math (HIP) ---------------------------------------------------------------
    def step(self, timestep: int, latents: torch.Tensor, model_output: torch.Tensor) -> torch.Tensor:
        coef = self.step_coefficients(timestep).reshape(1, 8).to(latents.device)
        out = latents.float().contiguous().clone()
        eps = model_output.float().contiguous()
        if eps.numel() != out.numel():
            raise ValueError(f"model_output {tuple(model_output.shape)} does not match latents {tuple(latents.shape)}")
        ops.cfg_ddpm_step(out.view(1, -1), eps.view(1, -1), None, coef, 0, 0.0, do_cfg=False, deterministic=True)
        return out

    def add_noise(self, original_samples: torch.Tensor, timesteps) -> torch.Tensor:
        a = np.float32(self.alpha_cumprod[int(timesteps)].item())
        noise = self.draw_noise(tuple(original_samples.shape)).to(original_samples.device)
        return ops.add_noise(original_samples.float(), noise, float(np.sqrt(a)), float(np.sqrt(np.float32(1) - a)))
