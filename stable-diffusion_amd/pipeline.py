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
    ``strength_semantics`` selects the img2img schedule slice, ``return_latents`` for tests, ``all_gather=True`` ends
    the call with the one RCCL all-gather of the decoded images over the ranks of a sharded job (every rank passes
    the same number of prompts), ``deterministic_plan=True`` plans every launch from ONE image's rows (ops.plan_samples):
    image i is then bit-identical whatever batch (or shard of a multi-GPU job) it runs in, at a throughput cost,
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


_TEMB_CACHE = {}


def time_embedding_table(timesteps):
    """[S, 320] f32 (CPU): get_time_embedding of every step of a schedule, kept per schedule (a few KB each)."""
    key = tuple(int(t) for t in timesteps)
    tab = _TEMB_CACHE.get(key)
    if tab is None:
        if len(_TEMB_CACHE) > 64:
            _TEMB_CACHE.clear()
        tab = _TEMB_CACHE[key] = torch.cat([get_time_embedding(t) for t in key])
    return tab


WHOLE_LOOP_GRAPH = True   # run(): all n_steps iterations as ONE hipGraph launch (False: one replay per step)


class DenoiseLoop:
    """The hot loop of generate() (pipeline.py:123-153) as hipGraphs.

    ``run()`` from step 0 launches ONE graph that holds all ``n_steps`` iterations of [UNet forward on 2B latents ->
    fused CFG + sampler step]; the schedule index of each iteration is baked into its nodes (time-bias row ``i`` of the
    table, coefficient row ``i``), so the whole-loop graph has no device-side counter and no gather.  ``step()``
    (teacher-forced tests, per-step preview, traces) replays a ONE-step graph = [time-bias row gather -> UNet ->
    CFG + step -> counter++] that reads its index from a device counter.  Both graphs share one memory pool.

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
        self.coef = self._coef_with_scale(coef, cfg_scale).to(dev).contiguous()
        self.host_step = 0            # host mirror of the device-side schedule index
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        unet = diffusion.unet
        self.kv = unet.prompt_state(context)
        temb = time_embedding_table(timesteps).to(dev)                                # [S,320], bit-exact table
        with ops.plan_samples(1):     # (rows = schedule steps, not samples: never planned per sample)
            self.tb_table = unet.time_bias_table(diffusion.time_embedding(temb))      # [S, sum(Cout)] f32
        self.tb_row = torch.empty(self.tb_table.shape[1], dtype=torch.float32, device=dev)
        self.tb = unet.time_bias_slices(self.tb_row)
        noise = _resolve_noise(noise)
        self.noise = None if noise is None else noise.to(dev, non_blocking=True).contiguous()
        self.graph = None
        self.full_graph = None
        if use_graph:
            self._capture()

    @staticmethod
    def _coef_with_scale(coef, cfg_scale):
        """The sampler's coefficient rows with the guidance scale in the spare column 7: the captured graphs read it
        from there (sdamd_cfg_ddpm_step with a NaN scale argument), so one capture serves every cfg_scale."""
        coef = coef.clone()
        coef[:, 7] = float(cfg_scale)
        return coef

    def _body_at(self, i):
        """Iteration ``i`` with its schedule index baked in (whole-loop graph): no counter, no row gather."""
        tb = self.diffusion.unet.time_bias_slices(self.tb_table[i])
        eps = self.diffusion.run(self.latents, self.kv, tb, copies=2 if self.do_cfg else 1)
        ops.cfg_ddpm_step(self.latents, eps, self.noise, self.coef, i, None, self.do_cfg, deterministic=self.noise is None)

    def _capture_full(self):
        """All n_steps iterations in one hipGraph (captured lazily by the first run() from step 0)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.graph.pool()):
            for i in range(self.n_steps):
                self._body_at(i)
        self.full_graph = g

    def _body(self):
        ops.gather_row(self.tb_row, self.tb_table, step_ptr=self.counter)
        # pipeline.py:147 `latents.repeat(2, 1, 1, 1)`: both CFG halves see the same latents, so the UNet's first
        # conv runs once and writes both copies (no [x; x] tensor is built)
        eps = self.diffusion.run(self.latents, self.kv, self.tb, copies=2 if self.do_cfg else 1)
        ops.cfg_ddpm_step(self.latents, eps, self.noise, self.coef, 0, None, self.do_cfg, step_ptr=self.counter,
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

    def rebind(self, context, timesteps, coef, noise, latents, cfg_scale=None):
        """Re-use the captured graph for another call of the same shape: new prompts, schedule rows, noise and start
        latents are copied into the buffers the graph reads.  Returns False (nothing changed) when the shapes differ."""
        if len(timesteps) != self.n_steps or tuple(latents.shape) != tuple(self.latents.shape) or \
                (noise is None) != (self.noise is None) or tuple(coef.shape) != tuple(self.coef.shape):
            return False
        if noise is None and bool((coef[:, 5] != 0).any()):
            return False
        self.set_context(context)
        temb = time_embedding_table(timesteps).to(self.latents.device)
        with ops.plan_samples(1):
            self.tb_table.copy_(self.diffusion.unet.time_bias_table(self.diffusion.time_embedding(temb)))
        if cfg_scale is not None:
            self.cfg_scale = float(cfg_scale)
        self.coef.copy_(self._coef_with_scale(coef, self.cfg_scale), non_blocking=True)
        self.reset(latents)
        if noise is not None:     # (possibly still being drawn: resolved last, after everything else is enqueued)
            self._noise_src = _resolve_noise(noise)      # the pinned source stays referenced until the next call
            self.noise.copy_(self._noise_src, non_blocking=True)
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
        if self.graph is not None and WHOLE_LOOP_GRAPH and trace is None and self.host_step == 0:
            if self.full_graph is None:
                self._capture_full()
            self.full_graph.replay()          # the whole schedule: one launch
            self.host_step = self.n_steps
            return self.latents
        for _ in range(self.host_step, self.n_steps):
            self.step()
            if trace is not None:
                trace.append(self.latents.clone())
        return self.latents


_LOOP_CACHE = {}      # generate(): captured DenoiseLoops by problem shape (graph capture is paid once per shape)
LOOP_CACHE_SIZE = 2   # each entry pins a private graph memory pool with a full batch of UNet activations (GBs at batch 8+)


def _weights_stamp(module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


def _denoise_loop(diffusion, context, timesteps, coef, noise, latents, cfg_scale, do_cfg, use_graph):
    """A DenoiseLoop for this call: a cached captured loop of the same shape / weights / guidance scale with its
    tables refreshed in place, or a new one (captured now, kept for the next call)."""
    if not use_graph:
        return DenoiseLoop(diffusion, context, timesteps, coef, noise, latents, cfg_scale, do_cfg, False)
    # (a captured graph keeps the f32 mode it was captured under; the guidance scale is NOT in the key: it rides in the
    #  coefficient table the graph reads)
    key = (id(diffusion), diffusion.compute_dtype, tuple(latents.shape), str(latents.device), len(timesteps), bool(do_cfg),
           noise is None, ops.get_f32_mode(), ops.PLAN_SAMPLES)
    ent = _LOOP_CACHE.get(key)
    stamp = _weights_stamp(diffusion)
    if ent is not None and ent[0] == stamp and ent[1].rebind(context, timesteps, coef, noise, latents, cfg_scale):
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


def decode_to_uint8(decoder, latents, micro_batch=8, deterministic_plan=False):
    """Final VAE decode + rescale/clamp/truncate -> uint8 [B,H,W,3] on the device."""
    outs = []
    for i in range(0, latents.shape[0], micro_batch):
        chunk = latents[i:i + micro_batch].contiguous()
        with ops.plan_samples(chunk.shape[0] if deterministic_plan else 0):
            outs.append(ops.image_to_uint8(decoder(chunk)))
    return outs[0] if len(outs) == 1 else torch.cat(outs)


_NOISE_POOL = None


def _noise_pool():
    global _NOISE_POOL
    if _NOISE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _NOISE_POOL = ThreadPoolExecutor(16, thread_name_prefix="sdamd-noise")
    return _NOISE_POOL


class PendingNoise:
    """Step noise still being drawn by the pool's threads: ``result()`` waits and returns the [S, B, ...] tensor.
    DenoiseLoop resolves it last, so the draws overlap the CLIP launches and the loop's table refresh."""

    def __init__(self, noise, futures):
        self.noise, self.futures = noise, futures

    def result(self):
        for f in self.futures:
            f.result()
        self.futures = []
        return self.noise


def _resolve_noise(noise):
    return noise.result() if isinstance(noise, PendingNoise) else noise


def _draw_step_noise(samplers, timesteps, shape, pending=False):
    """The per-step draws of ddpm.py:82 for every image: [S, B, *shape] f32 (zero rows where t == 0: no draw there), pinned
    when a HIP device is present so the upload is one asynchronous copy.

    Per image the draws are consecutive calls on that image's own CPU generator, so they are taken as ONE
    ``randn((n_draws, numel))``: torch's CPU normal_ fills a tensor of >= 16 elements in blocks of 16 uniforms (Box-Muller
    within the block), hence a tensor whose per-draw size is a multiple of 16 is bit-identical to the sequence of separate
    draws (tests/test_host_and_abi.py::test_batched_step_noise_equals_draw_by_draw pins this against draw-by-draw calls).
    The images have independent generators and are drawn on a pool thread each (randn releases the GIL);
    ``pending=True`` returns a PendingNoise at once (nothing else may touch the generators until it is resolved)."""
    S, B = len(timesteps), len(samplers)
    numel = int(np.prod(shape))
    noisy = [si for si, t in enumerate(timesteps) if int(t) > 0]
    noise = torch.zeros((S, B) + shape, dtype=torch.float32, pin_memory=torch.cuda.is_available())
    if not noisy:
        return noise
    cpu_gen = all(sm.generator.device.type == "cpu" for sm in samplers)
    if numel % 16 == 0 and cpu_gen:
        idx = torch.tensor(noisy)

        def draw(bi):
            buf = torch.randn((len(noisy), numel), generator=samplers[bi].generator)
            noise[idx, bi] = buf.view((len(noisy),) + shape)

        if B == 1 and not pending:
            draw(0)
            return noise
        futures = [_noise_pool().submit(draw, bi) for bi in range(B)]
        return PendingNoise(noise, futures) if pending else PendingNoise(noise, futures).result()
    for si in noisy:
        for bi, sm in enumerate(samplers):
            noise[si, bi] = sm.draw_noise(shape)
    return noise


def prepare_latents(seeds, sampler_cls, steps, shape, device, encoder=None, images_u8=None, strength=0.7,
                    strength_semantics="reference", pending_noise=False, deterministic_plan=False):
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
        chunks = []
        for i in range(0, len(samplers), mb):
            with ops.plan_samples(min(mb, len(samplers) - i) if deterministic_plan else 0):
                chunks.append(encoder(ops.uint8_to_image(images_u8[i:i + mb].contiguous()), enc_noise[i:i + mb].contiguous()))
        latents = torch.cat(chunks)
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
    # (last use of the generators: with pending_noise the draws run on pool threads while the caller goes on)
    noise = _draw_step_noise(samplers, timesteps, tuple(shape[1:]), pending_noise) if samplers[0].stochastic else None
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
    all_gather=False,
    deterministic_plan=False,
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
            p0 = next(m.parameters(), None)
            if p0 is None or p0.device != device and not (p0.is_cuda and device.index is None):
                m.to(device)        # (already resident: skipped, Module.to walks ~700 submodules per model)
            # the text encoder always runs its f32 kernels (0.03 % of the FLOPs): its error differs
            # between the two CFG branches and would be amplified by the guidance scale
            engine.set_compute_dtype(m, torch.float32 if name == "clip" else dtype)

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
            images_u8, strength, strength_semantics, pending_noise=True, deterministic_plan=deterministic_plan)
        # (the step noise is still being drawn on pool threads: tokenising, the CLIP launches and the loop's table refresh
        #  run meanwhile; DenoiseLoop waits for it last)
        pos = _token_tensor(tokenizer, prompts, device)
        ncopies = 2 if do_cfg else 1
        with ops.plan_samples(ncopies * B if deterministic_plan else 0):
            context = encode_prompts(models["clip"], pos, _token_tensor(tokenizer, negs, device) if do_cfg else None)

        with ops.plan_samples(ncopies * B if deterministic_plan else 0):
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
        images = decode_to_uint8(models["decoder"], final, deterministic_plan=deterministic_plan)
        if all_gather:   # multi-GPU job (distributed.py): this rank's images -> every rank's, rank order, ONE collective
            from . import distributed
            images = distributed.all_gather_images(images)
        out = images.cpu().numpy()
        if return_latents:
            return (out if batched else out[0]), final.clone()   # (the loop's own buffer is re-used by the next call)
        return out if batched else out[0]
