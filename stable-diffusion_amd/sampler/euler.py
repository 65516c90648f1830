"""Euler sampler behind the reference's ``EulerSampler`` interface (sampler/euler.py:4-100); SURVEY.md §8f rank 3.

Two modes, one fused HIP kernel (``sdamd_cfg_ddpm_step``, a coefficient row per step):

* ``corrected=False`` -- the reference class to the letter: scaled-linear betas (:13-16), ``set_steps`` leaves the
  timesteps ASCENDING (:26-29, no ``[::-1]``), and ``step`` (:54-77) weighs the predicted x0 by
  ``sqrt(abar_prev)(1-abar_t)/(1-abar_prev)`` and the sample by ``sqrt(1-abar_t)(1-abar_t)/(1-abar_prev)`` plus
  DDPM's posterior noise.  ``step`` is pinned bit for bit to the reference class wherever that is finite
  (tests/golden/euler_step.npz); a whole run is not: the first timestep is 0, whose previous timestep is
  negative, so ``1 - abar_prev = 0`` and both weights are infinite.  The reference's pipeline DOES import this class
  (pipeline.py:5) and resolves ``sampler="euler"`` to it (pipeline.py:74-84), so ``generate(sampler="euler")`` there
  runs the ascending schedule and returns non-finite latents from the first step on.  Here the literal class stays
  reachable as ``generate(sampler="euler_reference")`` (same non-finite result, for parity of behaviour).
* ``corrected=True`` (what ``generate(sampler="euler")`` builds HERE: a deliberate departure from the reference,
  whose ``"euler"`` has no finite run to be faithful to) -- the sampler the name promises: descending
  timesteps and one explicit Euler step per interval of the probability-flow ODE in sigma space,
  ``x/sqrt(abar)`` moving by ``(sigma_prev - sigma_t) * eps`` with ``sigma = sqrt((1-abar)/abar)``; written
  back in the sample's own scale that is ``sqrt(abar_prev) * x0 + sqrt(1 - abar_prev) * eps``: deterministic,
  no per-step noise, ``abar_prev = 1`` after the last step.  The reference holds nothing that pins this mode
  (**parity unpinned**); it is checked against its closed form in the oracle.
"""
import numpy as np
import torch

from .ddpm import DDPMSampler


class EulerSampler(DDPMSampler):
    def __init__(self, generator: torch.Generator, training_steps=1000, beta_start: float = 0.00085,
                 beta_end: float = 0.0120, corrected: bool = False):
        super().__init__(generator, training_steps, beta_start, beta_end)
        self.corrected = bool(corrected)
        self.stochastic = not self.corrected      # literal mode draws DDPM's posterior noise (euler.py:70-73)

    def set_steps(self, steps):
        self.steps = steps
        ratio = self.training_steps // steps
        ts = np.arange(0, steps) * ratio                                   # euler.py:26-29: ascending
        self.timesteps = torch.from_numpy((ts[::-1].copy() if self.corrected else ts).astype(np.int64))

    def step_coefficients(self, timestep) -> torch.Tensor:
        """f32[8] row of the fused step kernel: {sqrt(abar_t), sqrt(1-abar_t), c0, cx, sigma, has_noise, cx_on_eps, 0}."""
        f = np.float32
        t = int(timestep)
        prev_t = self._get_previous_timestep(t)
        a_t = f(self.alpha_cumprod[t].item())
        a_prev = f(self.alpha_cumprod[prev_t].item()) if prev_t >= 0 else f(1.0)
        if self.corrected:
            vals = [np.sqrt(a_t), np.sqrt(f(1) - a_t), np.sqrt(a_prev), np.sqrt(f(1) - a_prev), f(0), f(0), f(1), f(0)]
            return torch.from_numpy(np.asarray(vals, dtype=np.float32))
        with np.errstate(divide="ignore", invalid="ignore"):               # prev_t < 0: 1 - abar_prev = 0, as in the reference
            c0 = (np.sqrt(a_prev) * (f(1) - a_t)) / (f(1) - a_prev)       # euler.py:63
            cx = np.sqrt(f(1) - a_t) * (f(1) - a_t) / (f(1) - a_prev)     # euler.py:64
            sigma = f(0)
            if t > 0:                                                      # euler.py:36-47,70-73
                sigma = np.sqrt(np.maximum((f(1) - a_prev) / (f(1) - a_t) * (f(1) - a_t / a_prev), f(1e-20)))
        vals = [np.sqrt(a_t), np.sqrt(f(1) - a_t), c0, cx, sigma, f(1.0 if t > 0 else 0.0), f(0), f(0)]
        return torch.from_numpy(np.asarray(vals, dtype=np.float32))

    def _get_variance(self, timestep: int) -> torch.Tensor:
        return torch.tensor(0.0) if self.corrected else super()._get_variance(timestep)

    def step(self, timestep: int, latents: torch.Tensor, model_output: torch.Tensor) -> torch.Tensor:
        if not self.corrected:
            return super().step(timestep, latents, model_output)
        from .ddim import DDIMSampler
        return DDIMSampler.step(self, timestep, latents, model_output)     # same kernel call, this class's coefficients
