#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference/src); never on the GPU box.  It
imports the reference's modules unmodified, fills them with the hash weights of
``stable-diffusion_amd/synth.py`` (same state_dict keys), feeds hash/seeded inputs and
stores the outputs.  Fixtures are data only -- no reference source is copied.

    python tools/gen_golden.py [keys] [small] [models] [pipe3] [img2img] [pipe50] [converter] [ddim] [nocfg] [euler] [fp16] [img2img50] [pipe50b]
"""
import contextlib
import io
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
GOLD = os.path.join(ROOT, "tests", "golden")

# pipeline.py:7 imports IPython.display (not installed here): stub two no-op callables.
_ipy = types.ModuleType("IPython")
_disp = types.ModuleType("IPython.display")
_disp.clear_output = lambda *a, **k: None
_disp.display = lambda *a, **k: None
_ipy.display = _disp
sys.modules.setdefault("IPython", _ipy)
sys.modules.setdefault("IPython.display", _disp)

import sd_amd  # noqa: E402  (alias loader of the dash-named package)
from sd_amd import synth  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import attention as ref_attention  # noqa: E402
    import clip as ref_clip  # noqa: E402
    import decoder as ref_decoder  # noqa: E402
    import diffusion as ref_diffusion  # noqa: E402
    import encoder as ref_encoder  # noqa: E402
    import pipeline as ref_pipeline  # noqa: E402
    from sampler.ddpm import DDPMSampler as RefDDPM  # noqa: E402

torch.set_grad_enabled(False)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def hin(name, shape, scale=1.0):
    """Hash-generated input tensor (regenerated identically by the tests)."""
    return torch.from_numpy(synth.hash_uniform("input:" + name, int(np.prod(shape))).reshape(shape)) * scale


def save(name, **arrays):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KB", flush=True)


def gen_small():
    # --- sampler tables (bit-exact targets) ---
    s = RefDDPM(torch.Generator().manual_seed(0))
    s.set_steps(50)
    t50 = s.timesteps.clone()
    temb50 = torch.cat([ref_pipeline.get_time_embedding(t) for t in t50])
    s20 = RefDDPM(torch.Generator().manual_seed(0)); s20.set_steps(20)
    s99 = RefDDPM(torch.Generator().manual_seed(0)); s99.set_steps(50); s99.set_strength(0.99)
    s08 = RefDDPM(torch.Generator().manual_seed(0)); s08.set_steps(50); s08.set_strength(0.8)
    save("sampler_tables", alpha_cumprod=s.alpha_cumprod, betas=s.betas, timesteps50=t50,
         timesteps20=s20.timesteps, timesteps50_strength099=s99.timesteps,
         timesteps50_strength08_len=np.int64(len(s08.timesteps)), time_emb50=temb50)
    # --- DDPM step / add_noise on seeded inputs ---
    out = {}
    for steps, ts in ((50, (980, 500, 20, 0)), (20, (950, 0))):
        for t in ts:
            g = torch.Generator().manual_seed(1234 + t)
            sm = RefDDPM(g); sm.set_steps(steps)
            lat = hin(f"ddpm.lat.{t}", (1, 4, 16, 16), 2.0)
            eps = hin(f"ddpm.eps.{t}", (4, 16, 16), 1.5)
            out[f"step{steps}_{t}"] = sm.step(torch.tensor(t), lat, eps)
            if t > 0:  # the noise the reference drew (CPU randn differs by ulps across host ISAs)
                out[f"noise{steps}_{t}"] = torch.randn((4, 16, 16), generator=torch.Generator().manual_seed(1234 + t))
    g = torch.Generator().manual_seed(77)
    sm = RefDDPM(g); sm.set_steps(50)
    out["add_noise_780"] = sm.add_noise(hin("ddpm.an", (1, 4, 16, 16)), torch.tensor(780))
    out["add_noise_780_noise"] = torch.randn((1, 4, 16, 16), generator=torch.Generator().manual_seed(77))
    save("ddpm_step", **out)
    # --- attention ops ---
    out = {}
    for tag, heads, dim, seq, causal in (("h8d40", 8, 320, 64, False), ("h8d80", 8, 640, 48, False),
                                         ("h8d160", 8, 1280, 16, False), ("clip", 12, 768, 77, True),
                                         ("vae", 1, 512, 64, False)):
        bias_in = tag in ("clip", "vae")
        m = quiet(ref_attention.SelfAttention, heads, dim, in_proj_bias=bias_in)
        synth.load_hashed(m, f"op.self.{tag}.")
        out["self_" + tag] = m(hin(f"self.{tag}", (2, seq, dim)), causal_mask=causal)
    for tag, heads, dim, seq in (("h8d40", 8, 320, 64), ("h8d160", 8, 1280, 16)):
        m = quiet(ref_attention.CrossAttention, heads, dim, 768, in_proj_bias=False)
        synth.load_hashed(m, f"op.cross.{tag}.")
        out["cross_" + tag] = m(hin(f"cross.x.{tag}", (2, seq, dim)), hin(f"cross.y.{tag}", (2, 77, 768)))
    save("attention_ops", **out)
    # --- blocks ---
    out = {}
    m = synth.load_hashed(quiet(ref_diffusion.UNET_ResidualBlock, 64, 128), "blk.res.")
    out["unet_res_64_128"] = m(hin("blk.res.x", (2, 64, 8, 8)), hin("blk.res.t", (1, 1280)))
    m = synth.load_hashed(quiet(ref_diffusion.UNET_ResidualBlock, 320, 320), "blk.res2.")
    out["unet_res_320_320"] = m(hin("blk.res2.x", (2, 320, 8, 8)), hin("blk.res2.t", (1, 1280)))
    m = synth.load_hashed(quiet(ref_diffusion.UNET_AttentionBlock, 8, 40), "blk.attn.")
    out["unet_attn_8x40"] = m(hin("blk.attn.x", (2, 320, 8, 8)), hin("blk.attn.ctx", (2, 77, 768)))
    m = synth.load_hashed(quiet(ref_clip.CLIPLayer, 12, 768), "blk.clip.")
    out["clip_layer"] = m(hin("blk.clip.x", (1, 77, 768)))
    m = synth.load_hashed(quiet(ref_decoder.VAE_ResidualBlock, 128, 256), "blk.vres.")
    out["vae_res_128_256"] = m(hin("blk.vres.x", (1, 128, 16, 16)))
    m = synth.load_hashed(quiet(ref_decoder.VAE_AttentionBlock, 512), "blk.vattn.")
    out["vae_attn_512"] = m(hin("blk.vattn.x", (1, 512, 8, 8)))
    m = synth.load_hashed(quiet(ref_diffusion.Upsample, 64), "blk.up.")
    out["upsample_64"] = m(hin("blk.up.x", (1, 64, 8, 8)))
    save("blocks", **out)


_MODELS = {}


def ref_models():
    if not _MODELS:
        t0 = time.time()
        _MODELS["clip"] = synth.load_hashed(quiet(ref_clip.CLIP), "clip.")
        _MODELS["encoder"] = synth.load_hashed(quiet(ref_encoder.VAE_Encoder), "encoder.")
        _MODELS["decoder"] = synth.load_hashed(quiet(ref_decoder.VAE_Decoder), "decoder.")
        _MODELS["diffusion"] = synth.load_hashed(quiet(ref_diffusion.Diffusion), "diffusion.")
        print(f"  reference models built + hashed in {time.time() - t0:.0f}s", flush=True)
    return _MODELS


def gen_models():
    m = ref_models()
    ids = torch.tensor([synth.synthetic_token_ids(0), synth.EMPTY_PROMPT_IDS])
    ctx = m["clip"](ids)
    save("clip", tokens=ids, out=ctx)
    out = {}
    for hw in (16, 64):
        lat = hin(f"diff.lat.{hw}", (2, 4, hw, hw), 2.0)
        temb = ref_pipeline.get_time_embedding(980 if hw == 64 else 500)
        t0 = time.time()
        out[f"out{hw}"] = m["diffusion"](lat, ctx, temb)
        print(f"  Diffusion fwd b2@{hw}: {time.time() - t0:.1f}s", flush=True)
    save("diffusion", **out)
    out = {}
    full = m["decoder"](hin("dec.lat.64", (1, 4, 64, 64), 0.5))
    out["dec64_sub8"] = full[:, :, ::8, ::8]
    out["dec64_crop"] = full[:, :, 200:264, 200:264]
    out["dec64_mean_abs"] = full.abs().mean()
    out["dec8"] = m["decoder"](hin("dec.lat.8", (2, 4, 8, 8), 0.5))
    save("vae_decoder", **out)
    out = {}
    out["enc64"] = m["encoder"](hin("enc.img.64", (2, 3, 64, 64)), hin("enc.noise.64", (2, 4, 8, 8)))
    out["enc512"] = m["encoder"](hin("enc.img.512", (1, 3, 512, 512)), hin("enc.noise.512", (1, 4, 64, 64)))
    save("vae_encoder", **out)


class _Trace:
    """Record every DDPMSampler.step output without touching reference files."""

    def __enter__(self):
        self.latents = []
        self._orig = RefDDPM.step
        rec = self.latents

        def step(inner_self, t, lat, eps):
            out = self._orig(inner_self, t, lat, eps)
            rec.append(out.clone())
            return out

        RefDDPM.step = step
        return self

    def __exit__(self, *a):
        RefDDPM.step = self._orig


def gen_pipeline(steps, name, every=1, prompt="synthetic:0", seed=42, **kw):
    m = ref_models()
    t0 = time.time()
    with _Trace() as tr:
        img = quiet(ref_pipeline.generate, prompt, "", models=m, seed=seed, device="cpu",
                    sampler="ddpm", steps=steps, cfg_scale=7.5, tokenizer=synth.SyntheticTokenizer(), **kw)
    print(f"  reference generate({steps} steps) {time.time() - t0:.0f}s", flush=True)
    lat = torch.stack(tr.latents)
    idx = sorted(set(list(range(0, len(tr.latents), every)) + [len(tr.latents) - 1]))
    save(name, image=img, latents=lat[idx], latent_steps=np.asarray(idx, dtype=np.int64),
         n_steps=np.int64(len(tr.latents)))


def gen_keys():
    """state_dict key names + shapes of the reference modules: the weights ABI (loader.py:18-27)."""
    out = {}
    with torch.device("meta"):
        mods = {"diffusion": quiet(ref_diffusion.Diffusion), "clip": quiet(ref_clip.CLIP),
                "decoder": quiet(ref_decoder.VAE_Decoder), "encoder": quiet(ref_encoder.VAE_Encoder)}
    for name, m in mods.items():
        sd = m.state_dict()
        shapes = -np.ones((len(sd), 4), dtype=np.int64)
        for i, v in enumerate(sd.values()):
            shapes[i, :v.dim()] = list(v.shape)
        out[name + "_keys"] = np.array(list(sd.keys()))
        out[name + "_shapes"] = shapes
    save("state_dict_keys", **out)


def gen_ddim():
    """The reference's DDIMSampler class (sampler/ddim.py) and a 3-step generate() through it.  The
    reference pipeline forgot the import (pipeline.py:4-5): the class is put into the pipeline module's
    namespace from here -- no reference file is modified."""
    from sampler.ddim import DDIMSampler as RefDDIM
    s = RefDDIM(torch.Generator().manual_seed(0))
    t_default = s.timesteps.clone()
    s20 = RefDDIM(torch.Generator().manual_seed(0)); s20.set_steps(20)
    s08 = RefDDIM(torch.Generator().manual_seed(0)); s08.set_steps(50); s08.set_strength(0.8)
    out = dict(alpha_cumprod=s.alpha_cumprod, betas=s.betas, timesteps_default=t_default, timesteps20=s20.timesteps,
               timesteps50_strength08=s08.timesteps)
    for steps, ts in ((50, (980, 500, 20, 0)), (20, (950, 0))):
        for t in ts:
            sm = RefDDIM(torch.Generator().manual_seed(1)); sm.set_steps(steps)
            lat = hin(f"ddim.lat.{t}", (1, 4, 16, 16), 2.0)
            eps = hin(f"ddim.eps.{t}", (1, 4, 16, 16), 1.5)
            out[f"step{steps}_{t}"] = sm.step(t, lat, eps)
    sm = RefDDIM(torch.Generator().manual_seed(77)); sm.set_steps(50)
    out["add_noise_780"] = sm.add_noise(hin("ddim.an", (1, 4, 16, 16)), torch.tensor(780))
    out["add_noise_780_noise"] = torch.randn((1, 4, 16, 16), generator=torch.Generator().manual_seed(77))
    save("ddim_step", **out)
    # pipeline: trace DDIM's step like _Trace does for DDPM
    ref_pipeline.DDIMSampler = RefDDIM
    m = ref_models()
    rec, orig = [], RefDDIM.step

    def step(inner_self, t, lat, eps):
        o = orig(inner_self, t, lat, eps)
        rec.append(o.clone())
        return o

    RefDDIM.step = step
    try:
        t0 = time.time()
        img = quiet(ref_pipeline.generate, "synthetic:0", "", models=m, seed=42, device="cpu", sampler="ddim", steps=3,
                    cfg_scale=7.5, tokenizer=synth.SyntheticTokenizer())
        print(f"  reference generate(ddim, 3 steps) {time.time() - t0:.0f}s", flush=True)
    finally:
        RefDDIM.step = orig
        del ref_pipeline.DDIMSampler
    save("pipeline_txt2img_ddim_3step", image=img, latents=torch.stack(rec), latent_steps=np.arange(len(rec), dtype=np.int64),
         n_steps=np.int64(len(rec)))


def gen_euler():
    """The reference's EulerSampler class (sampler/euler.py): schedule tables and its step wherever that is finite
    (previous timestep >= 0).  At t = 0 -- the first timestep of its ascending schedule -- it divides by zero."""
    from sampler.euler import EulerSampler as RefEuler
    s = RefEuler(torch.Generator().manual_seed(0))
    out = dict(alpha_cumprod=s.alpha_cumprod, betas=s.betas, timesteps_default=s.timesteps.clone())
    s.set_steps(50)
    out["timesteps50"] = s.timesteps.clone()
    s20 = RefEuler(torch.Generator().manual_seed(0)); s20.set_steps(20)
    out["timesteps20"] = s20.timesteps
    for steps, ts in ((50, (980, 500, 20)), (20, (950, 50))):
        for t in ts:
            sm = RefEuler(torch.Generator().manual_seed(4321 + t)); sm.set_steps(steps)
            lat = hin(f"euler.lat.{t}", (1, 4, 16, 16), 2.0)
            eps = hin(f"euler.eps.{t}", (1, 4, 16, 16), 1.5)
            out[f"step{steps}_{t}"] = sm.step(t, lat, eps)
            out[f"noise{steps}_{t}"] = torch.randn((1, 4, 16, 16), generator=torch.Generator().manual_seed(4321 + t))
    sm = RefEuler(torch.Generator().manual_seed(5)); sm.set_steps(50)
    first = sm.step(0, hin("euler.lat.0", (1, 4, 16, 16), 2.0), hin("euler.eps.0", (1, 4, 16, 16), 1.5))
    out["step50_0_finite_fraction"] = np.float64(torch.isfinite(first).double().mean().item())
    save("euler_step", **out)


def gen_converter():
    """Run the reference's model_converter on a synthetic CompVis-layout checkpoint and record what
    it produced.  The checkpoint's key list is read out of the reference's text (the quoted
    ``original_model[...]`` subscripts, i.e. the file format it expects -- data, not code); tensors are
    tiny (shape [2, 3], rank-1 for biases) except the eight VAE attention conv weights whose reshape
    targets are literal (model_converter.py:1040-1045).  Values are hash_uniform("ckpt:" + key)."""
    import json
    import re
    import zlib
    import tempfile
    from safetensors.torch import save_file
    import model_converter as ref_conv
    text = open("/root/reference/src/model_converter.py").read()
    keys = list(dict.fromkeys(re.findall(r"original_model\['([^']+)'\]", text)))
    shapes = {}
    for k in keys:
        if re.search(r"mid\.attn_1\.(q|k|v|proj_out)\.weight$", k):
            shapes[k] = (512, 512, 1, 1)
        elif re.search(r"mid\.attn_1\.(q|k|v|proj_out)\.bias$", k):
            shapes[k] = (512,)
        else:
            shapes[k] = (3,) if k.endswith(".bias") else (2, 3)
    ckpt = {k: torch.from_numpy(synth.hash_uniform("ckpt:" + k, int(np.prod(s))).reshape(s)) for k, s in shapes.items()}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synthetic.safetensors")
        save_file(ckpt, path)
        conv = quiet(ref_conv.load_from_standard_weights, path, "cpu")
    out = {"source": {k: list(s) for k, s in shapes.items()}, "converted": {}}
    for model, sd in conv.items():
        out["converted"][model] = {k: {"shape": list(v.shape), "crc32": zlib.crc32(v.contiguous().numpy().tobytes())}
                                   for k, v in sd.items()}
    path = os.path.join(GOLD, "converter_map.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    n = sum(len(v) for v in out["converted"].values())
    print(f"  wrote converter_map.json  {len(keys)} source keys -> {n} converted keys  {os.path.getsize(path) / 1024:.0f} KB")


def main():
    what = set(sys.argv[1:]) or {"small", "models", "pipe3"}
    os.makedirs(GOLD, exist_ok=True)
    if "keys" in what:
        gen_keys()
    if "small" in what:
        gen_small()
    if "models" in what:
        gen_models()
    if "pipe3" in what:
        gen_pipeline(3, "pipeline_txt2img_3step")
    if "img2img" in what:
        from PIL import Image
        rng = np.random.default_rng(7)
        pil = Image.fromarray(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8))
        # literal reference semantics: start_step = 1000 - int(1000*0.998) = 2 -> 3 of 5 steps
        gen_pipeline(5, "pipeline_img2img_5step_s0998", input_image=pil, strength=0.998)
    if "ddim" in what:
        gen_ddim()
    if "euler" in what:
        gen_euler()
    if "converter" in what:
        gen_converter()
    if "nocfg" in what:      # pipeline.py:66-70,136-137,150-151: no guidance, context = the prompt alone
        gen_pipeline(2, "pipeline_txt2img_nocfg_2step", do_cfg=False)
    if "fp16" in what:       # pipeline.py:124-139: the reference's own reduced-precision mode (fp16 autocast), on CPU
        gen_pipeline(2, "pipeline_txt2img_fp16_2step", fp16_enabled=True)
        gen_pipeline(2, "pipeline_txt2img_2step")     # the same run in fp32: what the fp16 mode deviates from
    if "pipe50" in what:
        gen_pipeline(50, "pipeline_txt2img_50step", every=1)
    if "img2img50" in what:
        # BASELINE config 4's schedule (50 steps, "strength 0.8" as intended = the last 40 steps, from t = 780) out of the
        # UNMODIFIED reference: called with strength=0.99 its literal set_strength (ddpm.py:50-53) slices exactly
        # timesteps[10:].  Input image: uniform uint8 noise, seed 7.
        from PIL import Image
        pil = Image.fromarray(np.random.default_rng(7).integers(0, 256, (512, 512, 3), dtype=np.uint8))
        gen_pipeline(50, "pipeline_img2img_50step_s099", every=5, input_image=pil, strength=0.99)
    if "pipe50b" in what:    # a second seed / prompt for the 50-step free-running bf16 budget
        gen_pipeline(50, "pipeline_txt2img_50step_seed43", every=5, prompt="synthetic:1", seed=43)


if __name__ == "__main__":
    main()
