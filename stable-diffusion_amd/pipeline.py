"""``generate()`` with the reference's signature (pipeline.py:15-171) on the HIP path.

Differences a caller can observe (all documented in DESIGN.md):
  * only the final latents are decoded (the reference decodes and displays a preview every step,
    pipeline.py:156-164; the returned image is identical);
  * noise always comes from a CPU ``torch.Generator`` (mt19937) seeded with ``seed`` -- the stream
    the CPU reference consumes -- drawn up front and uploaded, so the whole step loop is one
    replayed hipGraph with no host work in it;
  * ``fp16_enabled=True`` selects the bf16 kernels (the reference's fp16 autocast has no MI355X
    counterpart here); ``idle_device`` is accepted and ignored (288 GB of HBM: nothing to offload);
  * extensions: ``prompt`` may be a list (batch of independent images, image i uses ``seed + i``),
    ``strength_semantics`` selects the img2img schedule slice, ``return_latents`` for tests,
    ``preview(step_index, timestep, uint8 images [B,H,W,3])`` restores the per-step decode as a callback.
"""
import numpy as np
import torch

from . import engine, ops
from .sampler.ddim import DDIMSampler  # noqa: F401  (resolved by name, see _resolve_sampler)
from .sampler.ddpm import DDPMSampler  # noqa: F401
from .sampler.euler import EulerSampler as _EulerSampler


def EulerSampler(generator):
    """What ``sampler="euler"`` resolves to HERE: the corrected Euler sampler.  A deliberate departure: the reference
    resolves "euler" to its literal class (pipeline.py:5,74-84), whose ascending schedule divides by zero on the first
    step and returns non-finite latents (sampler/euler.py; both modes are described there)."""
    return _EulerSampler(generator, corrected=True)


def Euler_referenceSampler(generator):
    """``sampler="euler_reference"``: the reference's EulerSampler to the letter (non-finite from its first step on)."""
    return _EulerSampler(generator, corrected=False)

WIDTH = 512
HEIGHT = 512
LATENTS_WIDTH = WIDTH // 8
LATENT_HEIGHT = HEIGHT // 8


def rescale(x, old_range, new_range, clamp=False):
    """pipeline.py:173-183 (host helper, in place like the reference)."""
    old_min, old_max = old_range
    new_min, new_max = new_range
    x -= old_min
    x *= (new_max - new_min) / (old_max - old_min)
    x += new_min
    if clamp:
        x = torch.clamp(x, new_min, new_max)
    return x


def get_time_embedding(timestep):
    """pipeline.py:186-190, same torch-CPU ops -> bit-exact table."""
    freqs = torch.pow(10000, -torch.arange(start=0, end=160, dtype=torch.float32) / 160)
    x = torch.tensor([timestep], dtype=torch.float32)[:, None] * freqs[None]
    return torch.cat([torch.cos(x), torch.sin(x)], dim=-1)


class DenoiseLoop:
    """The hot loop of generate(): 1 hipGraph = [time-bias row gather -> UNet forward on 2B latents
    -> fused CFG + DDPM step -> step counter++], replayed once per timestep.

    Everything that does not depend on the latents is hoisted out of the loop: prompt K|V
    projections, the time-embedding MLP and all 22 time projections for every step, the DDPM
    coefficient table and the pre-drawn noise.
    """

    def __init__(self, diffusion, context, timesteps, coef, noise, latents, cfg_scale, do_cfg=True, use_graph=True):
        dev = latents.device
        self.diffusion = diffusion
        self.B = latents.shape[0]
        self.do_cfg = do_cfg
        self.cfg_scale = float(cfg_scale)
        self.n_steps = len(timesteps)
        self.latents = latents.float().contiguous().clone()
        if noise is None and bool((coef[:, 5] != 0).any()):
            raise ops._lib.SdamdError("DenoiseLoop: the coefficient table has stochastic rows (has_noise != 0) but no "
                                      "pre-drawn noise was passed")
        self.coef = coef.to(dev).contiguous()
        self.noise = None if noise is None else noise.to(dev).contiguous()
        self.host_step = 0            # host mirror of the device-side schedule index
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        unet = diffusion.unet
        self.kv = unet.prompt_state(context)
        temb = torch.cat([get_time_embedding(t) for t in timesteps]).to(dev)          # [S,320], bit-exact table
        self.tb_table = unet.time_bias_table(diffusion.time_embedding(temb))          # [S, sum(Cout)] f32
        self.tb_row = torch.empty(self.tb_table.shape[1], dtype=torch.float32, device=dev)
        self.tb = unet.time_bias_slices(self.tb_row)
        self.graph = None
        if use_graph:
            self._capture()

    def _body(self):
        ops.gather_row(self.tb_row, self.tb_table, step_ptr=self.counter)
        # pipeline.py:147 `latents.repeat(2, 1, 1, 1)`: both CFG halves see the same latents, so the UNet's first
        # conv runs once and writes both copies (no [x; x] tensor is built)
        eps = self.diffusion.run(self.latents, self.kv, self.tb, copies=2 if self.do_cfg else 1)
        ops.cfg_ddpm_step(self.latents, eps, self.noise, self.coef, 0, self.cfg_scale, self.do_cfg, step_ptr=self.counter,
                          deterministic=self.noise is None)
        ops.counter_add(self.counter, 1)

    def _capture(self):
        ops.ensure_workspaces(self.latents.device, (2 if self.do_cfg else 1) * self.B)   # never allocated under capture
        keep = self.latents.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):     # warm-up: packs weights, sizes workspaces
            self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.latents.copy_(keep)
        self.counter.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        self.latents.copy_(keep)
        self.counter.zero_()
        self.host_step = 0

    def set_context(self, context):
        """New prompts for an already captured loop: recompute the K|V projections into the
        buffers the graph reads (addresses stay fixed)."""
        fresh = self.diffusion.unet.prompt_state(context)
        for key, buf in self.kv.items():
            buf.copy_(fresh[key])

    def rebind(self, context, timesteps, coef, noise, latents):
        """Re-use the captured graph for another call of the same shape: new prompts, schedule rows, noise and start
        latents are copied into the buffers the graph reads.  Returns False (nothing changed) when the shapes differ."""
        if len(timesteps) != self.n_steps or tuple(latents.shape) != tuple(self.latents.shape) or \
                (noise is None) != (self.noise is None) or tuple(coef.shape) != tuple(self.coef.shape):
            return False
        if noise is None and bool((coef[:, 5] != 0).any()):
            return False
        self.set_context(context)
        temb = torch.cat([get_time_embedding(t) for t in timesteps]).to(self.latents.device)
        self.tb_table.copy_(self.diffusion.unet.time_bias_table(self.diffusion.time_embedding(temb)))
        self.coef.copy_(coef, non_blocking=True)
        if noise is not None:
            self.noise.copy_(noise, non_blocking=True)
        self.reset(latents)
        return True

    def reset(self, latents, step=0):
        """Restart from ``latents`` at schedule index ``step`` (teacher-forced parity tests)."""
        if not 0 <= int(step) < self.n_steps:
            raise IndexError(f"DenoiseLoop.reset: step {step} outside the {self.n_steps}-step schedule")
        self.latents.copy_(latents)
        self.counter.fill_(int(step))
        self.host_step = int(step)

    def step(self):
        """One denoising step at the current schedule index.  The index lives on the device (graph replay); the
        host mirrors it and refuses to run past the schedule (the kernels would no-op, see sdamd_cfg_ddpm_step)."""
        if self.host_step >= self.n_steps:
            raise IndexError(f"DenoiseLoop.step: all {self.n_steps} steps of the schedule have run; reset() first")
        self.host_step += 1
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()

    def run(self, trace=None):
        for _ in range(self.host_step, self.n_steps):
            self.step()
            if trace is not None:
                trace.append(self.latents.clone())
        return self.latents


_LOOP_CACHE = {}      # generate(): captured DenoiseLoops by problem shape (graph capture is paid once per shape)
LOOP_CACHE_SIZE = 4


def _weights_stamp(module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


def _denoise_loop(diffusion, context, timesteps, coef, noise, latents, cfg_scale, do_cfg, use_graph):
    """A DenoiseLoop for this call: a cached captured loop of the same shape / weights / guidance scale with its
    tables refreshed in place, or a new one (captured now, kept for the next call)."""
    if not use_graph:
        return DenoiseLoop(diffusion, context, timesteps, coef, noise, latents, cfg_scale, do_cfg, False)
    key = (id(diffusion), diffusion.compute_dtype, tuple(latents.shape), str(latents.device), len(timesteps), bool(do_cfg),
           float(cfg_scale), noise is None, ops.get_f32_mode())   # (a captured graph keeps the f32 mode it was captured under)
    ent = _LOOP_CACHE.get(key)
    stamp = _weights_stamp(diffusion)
    if ent is not None and ent[0] == stamp and ent[1].rebind(context, timesteps, coef, noise, latents):
        _LOOP_CACHE[key] = _LOOP_CACHE.pop(key)          # most recently used last
        return ent[1]
    loop = DenoiseLoop(diffusion, context, timesteps, coef, noise, latents, cfg_scale, do_cfg, True)
    _LOOP_CACHE.pop(key, None)
    _LOOP_CACHE[key] = (stamp, loop)
    while len(_LOOP_CACHE) > LOOP_CACHE_SIZE:
        _LOOP_CACHE.pop(next(iter(_LOOP_CACHE)))
    return loop


def clear_loop_cache():
    _LOOP_CACHE.clear()


def _resolve_sampler(name):
    upper, capital = f"{name.upper()}Sampler", f"{name.capitalize()}Sampler"
    if upper in globals():
        return globals()[upper]
    if capital in globals():
        return globals()[capital]
    raise ValueError(f"Sampler '{name}' not found")


def _token_tensor(tokenizer, prompts, device):
    # pipeline.py:51-53 calls batch_encode_plus; transformers >= 5 dropped that alias of __call__
    encode = getattr(tokenizer, "batch_encode_plus", None) or tokenizer
    ids = encode(list(prompts), padding="max_length", max_length=77).input_ids
    return torch.tensor(ids, dtype=torch.long, device=device)


def encode_prompts(clip, pos_tokens, neg_tokens=None):
    """[pos_0..pos_B-1, neg_0..neg_B-1] context (pipeline.py:64 generalised to a batch)."""
    toks = pos_tokens if neg_tokens is None else torch.cat([pos_tokens, neg_tokens])
    return clip(toks)


def decode_to_uint8(decoder, latents, micro_batch=8):
    """Final VAE decode + rescale/clamp/truncate -> uint8 [B,H,W,3] on the device."""
    outs = []
    for i in range(0, latents.shape[0], micro_batch):
        outs.append(ops.image_to_uint8(decoder(latents[i:i + micro_batch].contiguous())))
    return outs[0] if len(outs) == 1 else torch.cat(outs)


def prepare_latents(seeds, sampler_cls, steps, shape, device, encoder=None, images_u8=None, strength=0.7,
                    strength_semantics="reference"):
    """Per image: its own CPU generator and the reference's RNG call order:
    txt2img: randn(latents) then one randn per step with t>0; img2img: randn(encoder noise),
    randn(add_noise), then the per-step draws (none for a deterministic sampler: noise is None).
    Returns (timesteps, coef, noise [S,B,...], latents)."""
    samplers = []
    for s in seeds:
        g = torch.Generator(device="cpu")
        if s is not None:
            g.manual_seed(int(s))
        sm = sampler_cls(g)
        sm.set_steps(steps)
        samplers.append(sm)
    if images_u8 is not None:
        enc_noise = torch.stack([sm.draw_noise(shape[1:]) for sm in samplers]).to(device)
        mb = 8      # micro-batch like the decoder: a 512^2 x 128-channel f32 trunk tensor is 134 MB per image
        latents = torch.cat([encoder(ops.uint8_to_image(images_u8[i:i + mb].contiguous()), enc_noise[i:i + mb].contiguous())
                             for i in range(0, len(samplers), mb)])
        for sm in samplers:
            (sm.set_strength if strength_semantics == "reference" else sm.set_strength_steps)(strength=strength)
        if len(samplers[0].timesteps) == 0:
            raise IndexError(
                f"strength={strength} leaves no timesteps with the reference's set_strength semantics "
                f"(sampler/ddpm.py:50-53, start_step {samplers[0].start_step} of {steps} steps); "
                "pass strength_semantics='intended'")
        t0 = samplers[0].timesteps[0]
        latents = torch.cat([sm.add_noise(latents[i:i + 1], t0) for i, sm in enumerate(samplers)])
    else:
        latents = torch.stack([sm.draw_noise(shape[1:]) for sm in samplers]).to(device)
    timesteps = samplers[0].timesteps
    coef = samplers[0].coefficient_table(timesteps)
    noise = None
    if samplers[0].stochastic:
        noise = torch.zeros((len(timesteps), len(seeds)) + tuple(shape[1:]), dtype=torch.float32)
        for si, t in enumerate(timesteps):
            if int(t) > 0:
                for bi, sm in enumerate(samplers):
                    noise[si, bi] = sm.draw_noise(shape[1:])
    return timesteps, coef, noise, latents


def generate(
    prompt,
    negative_prompt,
    input_image=None,
    strength=0.7,
    do_cfg=True,
    cfg_scale=7,
    sampler="ddpm",
    steps=20,
    models=(),
    seed=None,
    device=None,
    idle_device=None,
    tokenizer=None,
    fp16_enabled=False,
    *,
    strength_semantics="reference",
    compute_dtype=None,
    use_graph=True,
    return_latents=False,
    trace=None,
    preview=None,
):
    with torch.no_grad():
        if not (0 < strength <= 1):
            raise ValueError("Strength must be between 0 and 1")
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise ops._lib.SdamdError("sd_amd.pipeline.generate runs on the HIP device only (device='cuda'); no CPU fallback")
        batched = isinstance(prompt, (list, tuple))
        prompts = list(prompt) if batched else [prompt]
        B = len(prompts)
        negs = list(negative_prompt) if isinstance(negative_prompt, (list, tuple)) else [negative_prompt] * B
        dtype = compute_dtype or (torch.bfloat16 if fp16_enabled else torch.float32)
        for name, m in models.items():
            m.to(device)
            # the text encoder always runs its f32 kernels (0.03 % of the FLOPs): its error differs
            # between the two CFG branches and would be amplified by the guidance scale
            engine.set_compute_dtype(m, torch.float32 if name == "clip" else dtype)

        pos = _token_tensor(tokenizer, prompts, device)
        context = encode_prompts(models["clip"], pos, _token_tensor(tokenizer, negs, device) if do_cfg else None)

        sampler_cls = _resolve_sampler(sampler)
        seeds = [None if seed is None else seed + i for i in range(B)]
        shape = (1, 4, LATENT_HEIGHT, LATENTS_WIDTH)
        images_u8 = None
        if input_image is not None:
            imgs = list(input_image) if isinstance(input_image, (list, tuple)) else [input_image] * B
            arr = np.stack([np.array(im.resize((WIDTH, HEIGHT))) for im in imgs])
            images_u8 = torch.from_numpy(arr).to(device).contiguous()
        timesteps, coef, noise, latents = prepare_latents(
            seeds, sampler_cls, steps, shape, device, models.get("encoder") if input_image is not None else None,
            images_u8, strength, strength_semantics)

        loop = _denoise_loop(models["diffusion"], context, timesteps, coef, noise, latents, cfg_scale, do_cfg, use_graph)
        if preview is None:
            final = loop.run(trace)
        else:   # the reference's per-step preview (pipeline.py:156-164) as an opt-in callback
            for i in range(loop.n_steps):
                loop.step()
                if trace is not None:
                    trace.append(loop.latents.clone())
                preview(i, int(timesteps[i]), decode_to_uint8(models["decoder"], loop.latents.clone()).cpu().numpy())
            final = loop.latents
        images = decode_to_uint8(models["decoder"], final)
        out = images.cpu().numpy()
        if return_latents:
            return (out if batched else out[0]), final.clone()   # (the loop's own buffer is re-used by the next call)
        return out if batched else out[0]
