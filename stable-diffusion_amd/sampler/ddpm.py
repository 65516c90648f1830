"""DDPM ancestral sampler behind the reference's ``DDPMSampler`` interface (sampler/ddpm.py:4-111).

Schedules (betas, alpha_cumprod, timesteps) and the five per-step scalars are computed on the
host with the very torch-CPU ops the reference uses, so they match it bit for bit; tensor math
runs in the fused HIP kernel ``sdamd_cfg_ddpm_step`` (CFG combine + step in one launch).
Noise is always drawn from ``generator`` on the generator's own device (the CPU mt19937 stream
when the generator is a CPU generator -- what the CPU reference consumes) and uploaded.
"""
import numpy as np
import torch

from .. import ops


class DDPMSampler:
    stochastic = True       # one N(0,1) draw per step with t > 0 (ddpm.py:79-83)

    def __init__(self, generator: torch.Generator, training_steps=1000, beta_start: float = 0.00085,
                 beta_end: float = 0.0120):
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, training_steps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alpha_cumprod = torch.cumprod(self.alphas, 0)
        self.one = torch.tensor(1.0)
        self.generator = generator
        self.training_steps = training_steps
        self.steps = training_steps
        self.timesteps = torch.from_numpy(np.arange(0, training_steps)[::-1].copy())

    # ---- schedule (host, bit-exact with the reference) -----------------------------------
    def set_steps(self, steps):
        self.steps = steps
        ratio = self.training_steps // steps
        self.timesteps = torch.from_numpy((np.arange(0, steps) * ratio).round()[::-1].copy().astype(np.int64))

    def set_strength(self, strength=0.7):
        """Literal reference semantics (ddpm.py:50-53): the slice start is counted in TRAINING
        steps, so e.g. strength 0.8 with 50 inference steps leaves no timesteps at all."""
        start_step = self.training_steps - int(self.training_steps * strength)
        self.timesteps = self.timesteps[start_step:]
        self.start_step = start_step

    def set_strength_steps(self, strength=0.7):
        """Evidently intended semantics: skip ``steps - int(steps*strength)`` of
        the inference steps.  Extension, not in the reference."""
        start_step = self.steps - int(self.steps * strength)
        self.timesteps = self.timesteps[start_step:]
        self.start_step = start_step

    def _get_previous_timestep(self, timestep: int) -> int:
        return timestep - self.training_steps // self.steps

    def _get_variance(self, timestep: int) -> torch.Tensor:
        prev_t = self._get_previous_timestep(timestep)
        a_t = self.alpha_cumprod[timestep]
        a_prev = self.alpha_cumprod[prev_t] if prev_t >= 0 else self.one
        current_beta_t = 1 - a_t / a_prev
        variance = (1 - a_prev) / (1 - a_t) * current_beta_t
        return torch.clamp(variance, min=1e-20)

    def step_coefficients(self, timestep) -> torch.Tensor:
        """f32[8] = {sqrt(abar_t), sqrt(1-abar_t), c0, cx, sigma, has_noise, 0, 0}: the 0-dim
        tensors of ddpm.py:59-83, op for op, in numpy float32 scalar arithmetic.  (torch's scalar
        ``** 0.5`` / ``sqrt`` is not correctly rounded on every host CPU -- 1-ulp differences were
        measured between hosts -- whereas IEEE float32 here is portable and equals the reference's
        values on the fixture host.)"""
        f = np.float32
        t = int(timestep)
        prev_t = self._get_previous_timestep(t)
        a_t = f(self.alpha_cumprod[t].item())
        a_prev = f(self.alpha_cumprod[prev_t].item()) if prev_t >= 0 else f(1.0)
        beta_cum_t = f(1) - a_t
        beta_cum_prev = f(1) - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = f(1) - cur_alpha
        c0 = (np.sqrt(a_prev) * cur_beta) / beta_cum_t
        cx = np.sqrt(cur_alpha) * beta_cum_prev / beta_cum_t
        sigma = f(0)
        if t > 0:
            sigma = np.sqrt(np.maximum((f(1) - a_prev) / (f(1) - a_t) * (f(1) - a_t / a_prev), f(1e-20)))
        vals = [np.sqrt(a_t), np.sqrt(beta_cum_t), c0, cx, sigma, f(1.0 if t > 0 else 0.0), f(0), f(0)]
        return torch.from_numpy(np.asarray(vals, dtype=np.float32))

    def coefficient_table(self, timesteps=None) -> torch.Tensor:
        ts = self.timesteps if timesteps is None else timesteps
        return torch.stack([self.step_coefficients(t) for t in ts])

    def draw_noise(self, shape, dtype=torch.float32):
        return torch.randn(shape, generator=self.generator, device=self.generator.device, dtype=dtype)

    # ---- tensor math (HIP) ---------------------------------------------------------------
    def step(self, timestep: int, latents: torch.Tensor, model_output: torch.Tensor) -> torch.Tensor:
        """x_{t-1} from x_t and the (already CFG-combined) noise prediction; HIP tensors, f32."""
        coef = self.step_coefficients(timestep).reshape(1, 8).to(latents.device)
        noise = None
        if int(timestep) > 0:
            noise = self.draw_noise(tuple(model_output.shape)).to(latents.device).contiguous()
        out = latents.float().contiguous().clone()
        eps = model_output.float().contiguous()
        if eps.numel() != out.numel():
            raise ValueError(f"model_output {tuple(model_output.shape)} does not match latents {tuple(latents.shape)}")
        ops.cfg_ddpm_step(out.view(1, -1), eps.view(1, -1), noise, coef, 0, 0.0, do_cfg=False, deterministic=noise is None)   # t == 0: has_noise = 0
        return out

    def add_noise(self, original_samples: torch.Tensor, timesteps) -> torch.Tensor:
        a = np.float32(self.alpha_cumprod[int(timesteps)].item())
        noise = self.draw_noise(tuple(original_samples.shape)).to(original_samples.device)
        return ops.add_noise(original_samples.float(), noise, float(np.sqrt(a)), float(np.sqrt(np.float32(1) - a)))
