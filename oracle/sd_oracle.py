"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A functional torch-CPU fp32 restatement of the reference's denoising hot path
(KhoaDuong2005/Stable-Diffusion).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product package
(``stable-diffusion_amd/``) never does and fails loudly without its HIP library.

Parity pin: every function here is checked against the *imported reference itself*
(``tools/gen_golden.py`` ran /root/reference/src on CPU in the build container and
committed the outputs under ``tests/golden/``); see ``tests/test_oracle_golden.py``.
The tokenizer boundary (text -> ids) is "parity unpinned": the vocabulary files are not
shipped with the reference, so both sides are fed identical int64 ids.

All models are plain functions over a ``state_dict`` (same key names as the reference,
``loader.py:18-27``), each citing the reference lines it restates.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- ops


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _gn(sd, p, x, eps=1e-5):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    w = sd[p + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[p + ".bias"], 1e-5)


def attention_core(q, k, v, n_heads, causal=False):
    """softmax(mask(q k^T) / sqrt(d)) v -- attention.py:47-68 (mask before scale, :55-61)."""
    b, sq, c = q.shape
    skv = k.shape[1]
    d = c // n_heads
    q = q.view(b, sq, n_heads, d).transpose(1, 2)
    k = k.view(b, skv, n_heads, d).transpose(1, 2)
    v = v.view(b, skv, n_heads, d).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    if causal:
        w = w.masked_fill(torch.ones_like(w, dtype=torch.bool).triu(1), -torch.inf)
    w = w / math.sqrt(d)
    w = F.softmax(w, dim=-1)
    return (w @ v).transpose(1, 2).reshape(b, sq, c)


def self_attention(sd, p, x, n_heads, causal=False):
    """attention.py:25-75."""
    q, k, v = _lin(sd, p + ".in_proj", x).chunk(3, dim=-1)
    return _lin(sd, p + ".out_proj", attention_core(q, k, v, n_heads, causal))


def cross_attention(sd, p, x, y, n_heads):
    """attention.py:90-133."""
    q = _lin(sd, p + ".q_proj", x)
    k = _lin(sd, p + ".k_proj", y)
    v = _lin(sd, p + ".v_proj", y)
    return _lin(sd, p + ".out_proj", attention_core(q, k, v, n_heads))


# --------------------------------------------------------------------------- UNet


def unet_residual_block(sd, p, x, time):
    """diffusion.py:20-44 (GN eps 1e-5)."""
    h = _conv(sd, p + ".conv_feature", F.silu(_gn(sd, p + ".groupnorm_feature", x)))
    h = h + _lin(sd, p + ".linear_time", F.silu(time))[:, :, None, None]
    h = _conv(sd, p + ".conv_merged", F.silu(_gn(sd, p + ".groupnorm_merged", h)))
    if (p + ".residual_layer.weight") in sd:
        x = _conv(sd, p + ".residual_layer", x, padding=0)
    return h + x


def unet_attention_block(sd, p, x, ctx, n_heads=8):
    """diffusion.py:87-141 (GN eps 1e-6; GeGLU value = first half, gate = second, :128-130)."""
    n, c, h, w = x.shape
    long_res = x
    t = _conv(sd, p + ".conv_input", _gn(sd, p + ".groupnorm", x, 1e-6), padding=0)
    t = t.view(n, c, h * w).transpose(1, 2)
    t = t + self_attention(sd, p + ".attention_1", _ln(sd, p + ".layernorm_1", t), n_heads)
    t = t + cross_attention(sd, p + ".attention_2", _ln(sd, p + ".layernorm_2", t), ctx, n_heads)
    val, gate = _lin(sd, p + ".linear_geglu_1", _ln(sd, p + ".layernorm_3", t)).chunk(2, dim=-1)
    t = t + _lin(sd, p + ".linear_geglu_2", val * F.gelu(gate))
    t = t.transpose(1, 2).reshape(n, c, h, w)
    return _conv(sd, p + ".conv_output", t, padding=0) + long_res


# stage tables restating diffusion.py:194-262; entries: "c" plain conv, "d" stride-2 conv,
# "r" residual block, "a" attention block, "u" upsample+conv
UNET_ENCODERS = ["c", "ra", "ra", "d", "ra", "ra", "d", "ra", "ra", "d", "r", "r"]
UNET_BOTTLENECK = "rar"
UNET_DECODERS = ["r", "r", "ru", "ra", "ra", "rau", "ra", "ra", "rau", "ra", "ra", "ra"]


def _unet_stage(sd, p, kinds, x, ctx, time):
    for i, kind in enumerate(kinds):
        q = f"{p}.{i}"
        if kind == "c":
            x = _conv(sd, q, x)
        elif kind == "d":
            x = _conv(sd, q, x, stride=2)
        elif kind == "r":
            x = unet_residual_block(sd, q, x, time)
        elif kind == "a":
            x = unet_attention_block(sd, q, x, ctx)
        elif kind == "u":  # diffusion.py:148-150
            x = _conv(sd, q + ".conv", F.interpolate(x, scale_factor=2, mode="nearest"))
    return x


def unet_forward(sd, x, ctx, time, p="unet"):
    """diffusion.py:264-281; time is the [1,1280] TimeEmbedding output."""
    skips = []
    for i, kinds in enumerate(UNET_ENCODERS):
        x = _unet_stage(sd, f"{p}.encoders.{i}", kinds, x, ctx, time)
        skips.append(x)
    x = _unet_stage(sd, f"{p}.bottleneck", UNET_BOTTLENECK, x, ctx, time)
    for i, kinds in enumerate(UNET_DECODERS):
        x = torch.cat((x, skips.pop()), dim=1)
        x = _unet_stage(sd, f"{p}.decoders.{i}", kinds, x, ctx, time)
    return x


def time_mlp(sd, t320):
    """TimeEmbedding, diffusion.py:159-168."""
    return _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t320)))


def diffusion_forward(sd, latent, ctx, t320):
    """Diffusion.forward, diffusion.py:309-324 (+ UNET_Output_Layer :289-299)."""
    x = unet_forward(sd, latent, ctx, time_mlp(sd, t320))
    return _conv(sd, "final.conv", F.silu(_gn(sd, "final.groupnorm", x)))


# --------------------------------------------------------------------------- CLIP


def clip_forward(sd, tokens):
    """clip.py:86-97 (+ CLIPEmbedding :13-19, CLIPLayer :48-73, quick-GELU :67)."""
    x = sd["embedding.token_embedding.weight"][tokens.long()] + sd["embedding.position_embedding"]
    for i in range(12):
        p = f"layers.{i}"
        x = x + self_attention(sd, p + ".attention", _ln(sd, p + ".layernorm_1", x), 12, causal=True)
        h = _lin(sd, p + ".linear_1", _ln(sd, p + ".layernorm_2", x))
        x = x + _lin(sd, p + ".linear_2", h * torch.sigmoid(1.702 * h))
    return _ln(sd, "layernorm", x)


# --------------------------------------------------------------------------- VAE


def vae_residual_block(sd, p, x):
    """decoder.py:20-35."""
    h = _conv(sd, p + ".conv_1", F.silu(_gn(sd, p + ".groupnorm_1", x)))
    h = _conv(sd, p + ".conv_2", F.silu(_gn(sd, p + ".groupnorm_2", h)))
    if (p + ".residual_layer.weight") in sd:
        x = _conv(sd, p + ".residual_layer", x, padding=0)
    return h + x


def vae_attention_block(sd, p, x):
    """decoder.py:45-70 (1 head, d_head = C)."""
    n, c, h, w = x.shape
    t = _gn(sd, p + ".groupnorm", x).view(n, c, h * w).transpose(1, 2)
    t = self_attention(sd, p + ".attention", t, 1)
    return t.transpose(1, 2).reshape(n, c, h, w) + x


# decoder.py:75-124: index -> kind ("c1" 1x1 conv, "c" 3x3 conv, "r", "a", "up", "gn", "silu")
VAE_DECODER = ["c1", "c", "r", "a", "r", "r", "r", "r", "up", "c", "r", "r", "r", "up", "c",
               "r", "r", "r", "up", "c", "r", "r", "r", "gn", "silu", "c"]
# encoder.py:8-58 ("d" = stride-2 conv after F.pad(0,1,0,1), :65-66)
VAE_ENCODER = ["c", "r", "r", "d", "r", "r", "d", "r", "r", "d", "r", "r", "r", "a", "r",
               "gn", "silu", "c", "c1"]


def _vae_seq(sd, kinds, x):
    for i, kind in enumerate(kinds):
        p = str(i)
        if kind == "c1":
            x = _conv(sd, p, x, padding=0)
        elif kind == "c":
            x = _conv(sd, p, x)
        elif kind == "d":
            x = _conv(sd, p, F.pad(x, (0, 1, 0, 1)), stride=2, padding=0)
        elif kind == "r":
            x = vae_residual_block(sd, p, x)
        elif kind == "a":
            x = vae_attention_block(sd, p, x)
        elif kind == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif kind == "gn":
            x = _gn(sd, p, x)
        elif kind == "silu":
            x = F.silu(x)
    return x


def vae_decode(sd, latents):
    """VAE_Decoder.forward, decoder.py:126-134 (does not mutate the caller's tensor)."""
    return _vae_seq(sd, VAE_DECODER, latents / 0.18215)


def vae_encode(sd, image, noise):
    """VAE_Encoder.forward, encoder.py:60-86."""
    mean, logvar = torch.chunk(_vae_seq(sd, VAE_ENCODER, image), 2, dim=1)
    std = torch.clamp(logvar, -30, 20).exp().sqrt()
    return (mean + std * noise) * 0.18215


# --------------------------------------------------------------------------- sampler / pipeline


class DDPMSampler:
    """sampler/ddpm.py:4-111, same attribute names."""

    def __init__(self, generator, training_steps=1000, beta_start=0.00085, beta_end=0.0120):
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, training_steps,
                                    dtype=torch.float32) ** 2                      # :13
        self.alphas = 1.0 - self.betas
        self.alpha_cumprod = torch.cumprod(self.alphas, 0)                          # :16
        self.one = torch.tensor(1.0)
        self.generator = generator
        self.training_steps = training_steps
        self.timesteps = torch.from_numpy(np.arange(0, training_steps)[::-1].copy())

    def set_steps(self, steps):                                                     # :23-29
        self.steps = steps
        ratio = self.training_steps // steps
        self.timesteps = torch.from_numpy(
            (np.arange(0, steps) * ratio).round()[::-1].copy().astype(np.int64))

    def set_strength(self, strength=0.7):                                           # :50-53 (literal)
        self.start_step = self.training_steps - int(self.training_steps * strength)
        self.timesteps = self.timesteps[self.start_step:]

    def _prev(self, t):                                                             # :31-33
        return t - self.training_steps // self.steps

    def coefficients(self, t):
        """(sqrt(abar_t), sqrt(1-abar_t), c0, cx, sigma): the 0-dim fp32 values of :59-77, :35-48.

        Evaluated in numpy float32 scalar arithmetic, op for op as the reference writes them.
        torch's own scalar ``** 0.5`` / ``sqrt`` is NOT correctly rounded on every host (an AMD
        EPYC host returned sqrt values 1 ulp off an Intel host's); IEEE float32 arithmetic in
        numpy is, and reproduces the fixture host's reference values bit for bit."""
        f = np.float32
        t = int(t)
        pt = self._prev(t)
        a_t = f(self.alpha_cumprod[t].item())
        a_p = f(self.alpha_cumprod[pt].item()) if pt >= 0 else f(1.0)
        b_t, b_p = f(1) - a_t, f(1) - a_p
        cur_a = a_t / a_p
        cur_b = f(1) - cur_a
        c0 = (np.sqrt(a_p) * cur_b) / b_t
        cx = np.sqrt(cur_a) * b_p / b_t
        var = np.maximum((f(1) - a_p) / (f(1) - a_t) * (f(1) - a_t / a_p), f(1e-20))
        return tuple(torch.tensor(v, dtype=torch.float32) for v in (np.sqrt(a_t), np.sqrt(b_t), c0, cx, np.sqrt(var)))

    def step(self, t, latents, model_output, noise=None):                           # :55-87
        """``noise``: supply the N(0,1) draw instead of consuming the generator (torch's CPU randn
        differs by ulps between host ISAs, so fixtures carry the draw the reference made)."""
        sa, sb, c0, cx, sigma = self.coefficients(t)
        x0 = (latents - sb * model_output) / sa
        prev = x0 * c0 + cx * latents
        if int(t) > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=self.generator, dtype=model_output.dtype)
            prev = prev + sigma * noise
        return prev

    def add_noise(self, x, t, noise=None):                                          # :91-111
        a = np.float32(self.alpha_cumprod[int(t)].item())
        if noise is None:
            noise = torch.randn(x.shape, generator=self.generator, dtype=x.dtype)
        return float(np.sqrt(a)) * x + float(np.sqrt(np.float32(1) - a)) * noise


class DDIMSampler:
    """sampler/ddim.py:8-93 (eta = 0), same attribute names.  The reference's pipeline does not import
    it (pipeline.py:4-5); the goldens were made with the class injected into the pipeline module."""

    def __init__(self, generator, training_steps=1000):
        scale = 1000 / training_steps                                               # :15-20
        self.betas = torch.linspace(scale * 0.0001, scale * 0.02, training_steps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alpha_cumprod = torch.cumprod(self.alphas, 0)
        self.one = torch.tensor(1.0)
        self.generator = generator
        self.training_steps = training_steps
        self.set_steps(50)                                                          # :29-30

    def set_steps(self, steps):                                                     # :32-36
        self.steps = steps
        ratio = self.training_steps // steps
        self.timesteps = torch.from_numpy(
            (np.arange(0, steps) * ratio).round()[::-1].copy().astype(np.int64))

    def set_strength(self, strength=0.7):                                           # :42-45 (literal)
        self.start_step = self.training_steps - int(self.training_steps * strength)
        self.timesteps = self.timesteps[self.timesteps >= self.start_step]

    def _prev(self, t):                                                             # :38-40
        return max(int(t) - self.training_steps // self.steps, 0)

    def step(self, t, latents, model_output, noise=None):                           # :50-72
        f = np.float32
        a_t = f(self.alpha_cumprod[int(t)].item())
        a_p = f(self.alpha_cumprod[self._prev(t)].item())
        sa, sb, ca, cb = (torch.tensor(v, dtype=torch.float32) for v in
                          (np.sqrt(a_t), np.sqrt(f(1) - a_t), np.sqrt(a_p), np.sqrt(f(1) - a_p)))
        x0 = (latents - model_output * sb) / sa
        return ca * x0 + cb * model_output

    add_noise = DDPMSampler.add_noise                                               # :74-93, same math


class EulerSampler(DDPMSampler):
    """sampler/euler.py:4-100.  ``corrected=False``: the class as written -- ascending timesteps (:26-29) and the
    step of :54-77, finite only where the previous timestep is >= 0.  ``corrected=True``: descending timesteps and
    one explicit Euler step of the probability-flow ODE in sigma space per interval (no reference counterpart:
    parity unpinned; stated here in its sigma form, which the product's closed form has to reproduce)."""

    def __init__(self, generator, training_steps=1000, beta_start=0.00085, beta_end=0.0120, corrected=False):
        super().__init__(generator, training_steps, beta_start, beta_end)
        self.corrected = corrected

    def set_steps(self, steps):                                                     # :23-29 (no [::-1])
        self.steps = steps
        ts = (np.arange(0, steps) * (self.training_steps // steps)).astype(np.int64)
        self.timesteps = torch.from_numpy(ts[::-1].copy() if self.corrected else ts)

    def step(self, t, latents, model_output, noise=None):
        f = np.float32
        t = int(t)
        pt = self._prev(t)
        a_t = f(self.alpha_cumprod[t].item())
        a_p = f(self.alpha_cumprod[pt].item()) if pt >= 0 else f(1.0)
        if self.corrected:
            a_t, a_p = float(a_t), float(a_p)                                       # float64 closed form
            sig_t, sig_p = np.sqrt((1 - a_t) / a_t), np.sqrt((1 - a_p) / a_p)
            xs = latents.double() / np.sqrt(a_t)                                    # sample in sigma space
            xs = xs + (sig_p - sig_t) * model_output.double()                       # dx/dsigma = eps
            return (xs * np.sqrt(a_p)).float()
        with np.errstate(divide="ignore", invalid="ignore"):
            sa, sb = np.sqrt(a_t), np.sqrt(f(1) - a_t)
            c0 = (np.sqrt(a_p) * (f(1) - a_t)) / (f(1) - a_p)                       # :63
            cx = sb * (f(1) - a_t) / (f(1) - a_p)                                   # :64
            var = np.maximum((f(1) - a_p) / (f(1) - a_t) * (f(1) - a_t / a_p), f(1e-20))   # :36-47
        sa, sb, c0, cx, sigma = (torch.tensor(v, dtype=torch.float32) for v in (sa, sb, c0, cx, np.sqrt(var)))
        x0 = (latents - sb * model_output) / sa                                     # :61
        prev = c0 * x0 + cx * latents                                               # :66
        if t > 0:                                                                   # :70-73
            if noise is None:
                noise = torch.randn(model_output.shape, generator=self.generator, dtype=model_output.dtype)
            prev = prev + sigma * noise
        return prev


SAMPLERS = {"ddpm": DDPMSampler, "ddim": DDIMSampler, "euler": lambda g: EulerSampler(g, corrected=True)}


def get_time_embedding(t):
    """pipeline.py:186-190."""
    freqs = torch.pow(10000, -torch.arange(0, 160, dtype=torch.float32) / 160)
    x = torch.tensor([t], dtype=torch.float32)[:, None] * freqs[None]
    return torch.cat([torch.cos(x), torch.sin(x)], dim=-1)


def rescale(x, old, new, clamp=False):
    """pipeline.py:173-183 (out of place)."""
    x = (x - old[0]) * ((new[1] - new[0]) / (old[1] - old[0])) + new[0]
    return x.clamp(new[0], new[1]) if clamp else x


def generate(models, pos_ids, neg_ids, seed, steps=50, cfg_scale=7.5, input_image=None,
             strength_start=None, preview_every_step=False, latent_hw=(64, 64), trace=None, sampler="ddpm", do_cfg=True):
    """txt2img / img2img restatement of pipeline.generate (pipeline.py:15-171), batch 1.

    models: dict of state_dicts {"clip","diffusion","decoder","encoder"}.  Token ids go in
    directly (tokenizer boundary unpinned).  The 49 non-final preview decodes (:156-164) do
    not change the returned image and are skipped unless ``preview_every_step``.
    ``strength_start``: index into the step schedule to start from (img2img; the reference's
    ``set_strength`` slices with a training-step count, ddpm.py:51-52 -- see DESIGN.md).
    ``trace``: optional list that receives each step's latents.
    """
    g = torch.Generator().manual_seed(seed)
    if do_cfg:
        ctx = torch.cat([clip_forward(models["clip"], torch.tensor([pos_ids])),
                         clip_forward(models["clip"], torch.tensor([neg_ids]))])    # :64 [pos; neg]
    else:
        ctx = clip_forward(models["clip"], torch.tensor([pos_ids]))                 # :66-70
    sampler = SAMPLERS[sampler](g)
    sampler.set_steps(steps)
    shape = (1, 4, latent_hw[0], latent_hw[1])
    if input_image is not None:                                                     # :95-111
        img = rescale(torch.as_tensor(np.asarray(input_image), dtype=torch.float32), (0, 255), (-1, 1))
        img = img.unsqueeze(0).permute(0, 3, 1, 2)
        enc_noise = torch.randn(shape, generator=g)
        latents = vae_encode(models["encoder"], img, enc_noise)
        sampler.timesteps = sampler.timesteps[strength_start:]
        latents = sampler.add_noise(latents, sampler.timesteps[0])
    else:
        latents = torch.randn(shape, generator=g)                                   # :113
    image = None
    for t in sampler.timesteps:
        temb = get_time_embedding(t)
        if do_cfg:
            out = diffusion_forward(models["diffusion"], latents.repeat(2, 1, 1, 1), ctx, temb)
            eps = cfg_scale * (out[0:1] - out[1:2]) + out[1:2]                      # :149
        else:
            eps = diffusion_forward(models["diffusion"], latents, ctx, temb)        # :151
        latents = sampler.step(t, latents, eps[0])                                  # noise shape (4,h,w)
        if trace is not None:
            trace.append(latents.clone())
        if preview_every_step:
            image = to_uint8_image(vae_decode(models["decoder"], latents))
    if image is None:
        image = to_uint8_image(vae_decode(models["decoder"], latents))
    return image


def to_uint8_image(x):
    """pipeline.py:158-160: clamp to [0,255] then truncating cast, NCHW -> HWC of image 0."""
    x = rescale(x, (-1, 1), (0, 255), clamp=True)
    return x.permute(0, 2, 3, 1).to(torch.uint8).numpy()[0]
