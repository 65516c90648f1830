"""GPU parity of the product modules (reference-named classes executing on libsdamd) against the
golden outputs of the reference itself (tests/golden, made by tools/gen_golden.py) and against
the CPU oracle.  Tolerances are north_star's: relative L2 <= 1e-4 in fp32, <= 2e-2 in bf16.
"""
import numpy as np
import pytest
import torch

import sd_amd
from sd_amd import attention, clip, decoder, diffusion, encoder, engine, ops, pipeline, synth
from oracle import sd_oracle as O
from conftest import golden, rel_l2, hin, cpu_state_dicts

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
FP32, BF16 = torch.float32, torch.bfloat16
TOL = {FP32: 1e-4, BF16: 2e-2}
DTYPES = [FP32, BF16]


def build(cls, prefix, *args, **kw):
    m = synth.load_hashed(cls(*args, **kw), prefix)
    return m.to(DEV)


def with_dtype(m, dtype):
    return engine.set_compute_dtype(m, dtype)


# ------------------------------------------------------------------ attention modules
@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_modules_vs_reference(dtype):
    g = golden("attention_ops")
    for tag, heads, dim, seq, causal in (("h8d40", 8, 320, 64, False), ("h8d80", 8, 640, 48, False),
                                         ("h8d160", 8, 1280, 16, False), ("clip", 12, 768, 77, True),
                                         ("vae", 1, 512, 64, False)):
        m = with_dtype(build(attention.SelfAttention, f"op.self.{tag}.", heads, dim, in_proj_bias=tag in ("clip", "vae")), dtype)
        out = m(hin(f"self.{tag}", (2, seq, dim)).to(dtype).to(DEV), causal_mask=causal)
        assert rel_l2(out.float().cpu(), g["self_" + tag]) < TOL[dtype], tag
    for tag, heads, dim, seq in (("h8d40", 8, 320, 64), ("h8d160", 8, 1280, 16)):
        m = with_dtype(build(attention.CrossAttention, f"op.cross.{tag}.", heads, dim, 768, in_proj_bias=False), dtype)
        out = m(hin(f"cross.x.{tag}", (2, seq, dim)).to(dtype).to(DEV), hin(f"cross.y.{tag}", (2, 77, 768)).to(DEV))
        assert rel_l2(out.float().cpu(), g["cross_" + tag]) < TOL[dtype], tag


def test_flash_aliases_and_attention_type_strings():
    from sd_amd import flash_attention
    assert flash_attention.FlashSelfAttention is attention.SelfAttention
    assert flash_attention.FlashCrossAttention is attention.CrossAttention
    with pytest.raises(ValueError, match="Unknown attention type"):
        diffusion.UNET_AttentionBlock(8, 40, attention_type="cuda")


# ------------------------------------------------------------------ blocks
@pytest.mark.parametrize("dtype", DTYPES)
def test_blocks_vs_reference(dtype):
    g = golden("blocks")
    d = lambda name, shape: hin(name, shape).to(DEV)
    m = with_dtype(build(diffusion.UNET_ResidualBlock, "blk.res.", 64, 128), dtype)
    assert rel_l2(m(d("blk.res.x", (2, 64, 8, 8)), d("blk.res.t", (1, 1280))).cpu(), g["unet_res_64_128"]) < TOL[dtype]
    m = with_dtype(build(diffusion.UNET_ResidualBlock, "blk.res2.", 320, 320), dtype)
    assert rel_l2(m(d("blk.res2.x", (2, 320, 8, 8)), d("blk.res2.t", (1, 1280))).cpu(), g["unet_res_320_320"]) < TOL[dtype]
    m = with_dtype(build(diffusion.UNET_AttentionBlock, "blk.attn.", 8, 40), dtype)
    assert rel_l2(m(d("blk.attn.x", (2, 320, 8, 8)), d("blk.attn.ctx", (2, 77, 768))).cpu(), g["unet_attn_8x40"]) < TOL[dtype]
    m = with_dtype(build(clip.CLIPLayer, "blk.clip.", 12, 768), dtype)
    assert rel_l2(m(d("blk.clip.x", (1, 77, 768))).float().cpu(), g["clip_layer"]) < TOL[dtype]
    m = with_dtype(build(decoder.VAE_ResidualBlock, "blk.vres.", 128, 256), dtype)
    assert rel_l2(m(d("blk.vres.x", (1, 128, 16, 16))).cpu(), g["vae_res_128_256"]) < TOL[dtype]
    m = with_dtype(build(decoder.VAE_AttentionBlock, "blk.vattn.", 512), dtype)
    assert rel_l2(m(d("blk.vattn.x", (1, 512, 8, 8))).cpu(), g["vae_attn_512"]) < TOL[dtype]
    m = with_dtype(build(diffusion.Upsample, "blk.up.", 64), dtype)
    assert rel_l2(m(d("blk.up.x", (1, 64, 8, 8))).cpu(), g["upsample_64"]) < TOL[dtype]


# ------------------------------------------------------------------ whole models
@pytest.mark.parametrize("dtype", DTYPES)
def test_clip_vs_reference(models, dtype):
    g = golden("clip")
    out = with_dtype(models["clip"], dtype)(torch.from_numpy(g["tokens"]).to(DEV))
    assert out.dtype == dtype and tuple(out.shape) == (2, 77, 768)
    assert rel_l2(out.float().cpu(), g["out"]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw,t", [(16, 500), (64, 980)])
def test_diffusion_vs_reference(models, dtype, hw, t):
    """BASELINE config 2 (hw=64): UNET-only denoising step, latent 64x64x4, batch 2 (CFG)."""
    g = golden("diffusion")
    ctx = torch.from_numpy(golden("clip")["out"]).to(DEV)
    out = with_dtype(models["diffusion"], dtype)(hin(f"diff.lat.{hw}", (2, 4, hw, hw), 2.0).to(DEV), ctx,
                                                  pipeline.get_time_embedding(t).to(DEV))
    assert out.dtype == torch.float32 and tuple(out.shape) == (2, 4, hw, hw)
    assert rel_l2(out.cpu(), g[f"out{hw}"]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_unet_forward_signature(models, dtype):
    """UNET.forward(x, prompt, time[1,1280]) -> [B,320,h,w], checked against the CPU oracle."""
    dm = with_dtype(models["diffusion"], dtype)
    x, ctx, t = hin("unet.x", (2, 4, 8, 8), 2.0), hin("unet.ctx", (2, 77, 768)), hin("unet.t", (1, 1280))
    sd = {k: v.cpu() for k, v in dm.state_dict().items()}
    want = O.unet_forward(sd, x, ctx, t)
    got = dm.unet(x.to(DEV), ctx.to(DEV), t.to(DEV))
    assert tuple(got.shape) == (2, 320, 8, 8) and rel_l2(got.cpu(), want) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_vae_decoder_vs_reference(models, dtype):
    g = golden("vae_decoder")
    dec = with_dtype(models["decoder"], dtype)
    lat = hin("dec.lat.8", (2, 4, 8, 8), 0.5).to(DEV)
    keep = lat.clone()
    out = dec(lat)
    assert torch.equal(lat, keep), "decoder must not modify the caller's latents"
    assert rel_l2(out.cpu(), g["dec8"]) < TOL[dtype]
    full = dec(hin("dec.lat.64", (1, 4, 64, 64), 0.5).to(DEV)).cpu()
    assert tuple(full.shape) == (1, 3, 512, 512)
    assert rel_l2(full[:, :, ::8, ::8], g["dec64_sub8"]) < TOL[dtype]
    assert rel_l2(full[:, :, 200:264, 200:264], g["dec64_crop"]) < TOL[dtype]
    assert abs(float(full.abs().mean()) - float(g["dec64_mean_abs"])) < TOL[dtype] * float(g["dec64_mean_abs"])


@pytest.mark.parametrize("dtype", DTYPES)
def test_vae_encoder_vs_reference(models, dtype):
    g = golden("vae_encoder")
    enc = with_dtype(models["encoder"], dtype)
    out = enc(hin("enc.img.64", (2, 3, 64, 64)).to(DEV), hin("enc.noise.64", (2, 4, 8, 8)).to(DEV))
    assert rel_l2(out.cpu(), g["enc64"]) < TOL[dtype]
    out = enc(hin("enc.img.512", (1, 3, 512, 512)).to(DEV), hin("enc.noise.512", (1, 4, 64, 64)).to(DEV))
    assert rel_l2(out.cpu(), g["enc512"]) < TOL[dtype]


def test_state_dict_keys_are_the_reference_abi(models):
    keys = golden("state_dict_keys")
    for name in ("diffusion", "clip", "decoder", "encoder"):
        sd = models[name].state_dict()
        want = dict(zip(keys[name + "_keys"].tolist(), [tuple(s[s >= 0]) for s in keys[name + "_shapes"]]))
        assert set(sd) == set(want), name
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(int(x) for x in want[k]), (name, k)


# ------------------------------------------------------------------ pipeline.generate
def _gen(models, dtype, sampler="ddpm", **kw):
    for m in models.values():
        with_dtype(m, dtype)
    trace = []
    img, final = pipeline.generate("synthetic:0", "", models=models, seed=42, device="cuda", sampler=sampler, cfg_scale=7.5,
                                   tokenizer=synth.SyntheticTokenizer(), compute_dtype=dtype, trace=trace,
                                   return_latents=True, **kw)
    return img, torch.stack([t.cpu() for t in trace]), final


def _check_final_image(models, img, final_latents, g, dtype):
    """Final image parity.  The float image (before the truncating uint8 cast) is compared with the
    CPU oracle's decode of the REFERENCE's final latents at north_star's tolerance; the uint8 image
    is compared with the reference's uint8 image pixel by pixel (a 1e-6 wobble next to an integer
    boundary flips that pixel by one level, so fp32 allows +-1 on <1% of the pixels)."""
    ref_img = g["image"]
    assert img.shape == ref_img.shape and img.dtype == np.uint8
    sd = {k: v.cpu() for k, v in models["decoder"].state_dict().items()}
    want = O.vae_decode(sd, torch.from_numpy(g["latents"][-1]))
    odiff = np.abs(O.to_uint8_image(want).astype(np.int32) - ref_img.astype(np.int32))
    assert odiff.max() <= 1 and (odiff > 0).mean() < 0.01            # oracle reproduces the reference image
    # (bit-equal on the fixture host; another host's CPU conv rounding may flip a pixel by one level)
    got = with_dtype(models["decoder"], dtype)(final_latents).cpu()
    assert rel_l2(got, want) < TOL[dtype]
    diff = np.abs(img.astype(np.int32) - ref_img.astype(np.int32))
    if dtype == FP32:
        assert diff.max() <= 1 and (diff > 0).mean() < 0.01, (diff.max(), (diff > 0).mean())
    else:
        assert rel_l2(img.astype(np.float32), ref_img.astype(np.float32)) < TOL[dtype]


def test_fp32_weight_planes_are_made_once_and_exact_mode_matches(models):
    """fp32 Diffusion.forward (BASELINE config 2): the bf16x3 weight planes (ops.weight_planes) are made on the first
    call only -- every GEMM / conv weight the model hands to the kernels is a cached, prepared tensor, never a per-call
    temporary -- and the exact f32-MFMA mode gives the same output to f32-rounding level (both within the fp32 budget
    of the reference golden)."""
    g = golden("diffusion")
    ctx = torch.from_numpy(golden("clip")["out"]).to(DEV)
    lat, temb = hin("diff.lat.64", (2, 4, 64, 64), 2.0).to(DEV), pipeline.get_time_embedding(980).to(DEV)
    dm = with_dtype(models["diffusion"], torch.float32)
    out = dm(lat, ctx, temb)
    n = len(ops._W3)
    assert n > 100                                   # the UNet's Linear / conv weights
    out2 = dm(lat, ctx, temb)
    assert len(ops._W3) == n and torch.equal(out, out2)
    old = ops.set_f32_mode("exact")
    try:
        exact = dm(lat, ctx, temb)
    finally:
        ops.set_f32_mode(old)
    assert rel_l2(out.cpu(), g["out64"]) < TOL[torch.float32] and rel_l2(exact.cpu(), g["out64"]) < TOL[torch.float32]
    assert rel_l2(out.cpu(), exact.cpu()) < 5e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_cfg_shared_prefix_equals_the_duplicated_batch(models, dtype, monkeypatch):
    """UNET.run(copies=2) (the classifier-free-guidance batch [latents; latents], pipeline.py:147) computes the first
    residual block and the first attention block up to its cross attention once (diffusion.SHARE_CFG_PREFIX): same
    result as running the duplicated batch, to rounding (fp32: summation order of different tile plans only)."""
    dm = with_dtype(models["diffusion"], dtype)
    ctx = torch.from_numpy(golden("clip")["out"]).to(DEV)                      # [2, 77, 768]: two different prompts
    lat = hin("diff.lat.64", (2, 4, 64, 64), 2.0)[:1].to(DEV).contiguous()
    kv = dm.unet.prompt_state(ctx)
    tb = dm.unet.time_bias_slices(dm.unet.time_bias_table(dm.time_embedding(pipeline.get_time_embedding(500).to(DEV))).reshape(-1))
    outs = {}
    for share in (False, True):
        monkeypatch.setattr(diffusion, "SHARE_CFG_PREFIX", share)
        outs[share] = dm.run(lat, kv, tb, copies=2)
    assert tuple(outs[True].shape) == (2, 4, 64, 64)
    full = dm(torch.cat([lat, lat]), ctx, pipeline.get_time_embedding(500).to(DEV))   # the reference-signature forward: no sharing
    assert rel_l2(outs[False].cpu(), full.cpu()) < (1e-6 if dtype == torch.float32 else 1e-3)
    assert rel_l2(outs[True].cpu(), outs[False].cpu()) < (1e-5 if dtype == torch.float32 else 6e-3)
    assert not torch.equal(outs[True][0], outs[True][1])                       # the two prompts do differ


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_txt2img_3step_vs_reference(models, dtype):
    g = golden("pipeline_txt2img_3step")
    img, lat, final = _gen(models, dtype, steps=3)
    errs = [rel_l2(lat[i], g["latents"][i]) for i in range(3)]
    print(f"\n[{dtype}] 3-step free-running rel-L2: " + " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < TOL[dtype], errs
    _check_final_image(models, img, final, g, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_txt2img_ddim_3step_vs_reference(models, dtype):
    """sampler="ddim" (SURVEY §8f-3): the reference's DDIMSampler driven by its own pipeline loop."""
    g = golden("pipeline_txt2img_ddim_3step")
    img, lat, final = _gen(models, dtype, sampler="ddim", steps=3)
    errs = [rel_l2(lat[i], g["latents"][i]) for i in range(3)]
    print(f"\n[{dtype}] DDIM 3-step free-running rel-L2: " + " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < TOL[dtype], errs
    _check_final_image(models, img, final, g, dtype)


def test_generate_reduced_precision_vs_reference_fp16_mode(models):
    """fp16_enabled=True (pipeline.py:124-139) selects the bf16 mode.  Against the reference's own
    fp16-autocast run of the same 2 steps the bar is 3e-2 (the 2e-2 reduced-precision bar and the reference's own
    fp16-vs-fp32 deviation of ~1.5e-2 in quadrature); against the reference's fp32 run the mode is held to 2e-2 and
    has to be no further from it than the reference's fp16 mode is."""
    g16, g32 = golden("pipeline_txt2img_fp16_2step"), golden("pipeline_txt2img_2step")
    for m in models.values():
        with_dtype(m, FP32)
    trace = []
    img = pipeline.generate("synthetic:0", "", models=models, seed=42, device="cuda", sampler="ddpm", steps=2, cfg_scale=7.5,
                            tokenizer=synth.SyntheticTokenizer(), fp16_enabled=True, trace=trace)
    lat = torch.stack([t.cpu() for t in trace])
    for i in range(2):
        e16, e32 = rel_l2(lat[i], g16["latents"][i]), rel_l2(lat[i], g32["latents"][i])
        ref_dev = rel_l2(g16["latents"][i], g32["latents"][i])
        print(f"\nstep {i}: vs reference fp16 {e16:.1e}, vs reference fp32 {e32:.1e}; reference fp16 vs fp32 {ref_dev:.1e}")
        assert e16 < 3e-2 and e32 < TOL[BF16] and e32 <= ref_dev, (i, e16, e32, ref_dev)
    assert rel_l2(img.astype(np.float32), g16["image"].astype(np.float32)) < 3e-2


def test_generate_euler_vs_oracle(models):
    """sampler="euler"  builds the corrected Euler sampler: deterministic, descending schedule.
    **Parity unpinned** -- the reference's class has no finite run (sampler/euler.py, see sd_amd/sampler/euler.py);
    checked against the CPU oracle's sigma-space Euler loop over the oracle's own UNet, f32, 2 steps."""
    img, lat, final = _gen(models, FP32, sampler="euler", steps=2)
    sds = {name: {k: v.cpu() for k, v in m.state_dict().items()} for name, m in models.items()}
    trace = []
    want = O.generate(sds, synth.synthetic_token_ids(0), synth.EMPTY_PROMPT_IDS, seed=42, steps=2, cfg_scale=7.5,
                      trace=trace, sampler="euler")
    errs = [rel_l2(lat[i], trace[i]) for i in range(2)]
    assert torch.isfinite(final).all() and max(errs) < TOL[FP32], errs
    diff = np.abs(img.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_without_cfg_vs_reference(models, dtype):
    """do_cfg=False (pipeline.py:66-70,150-151): UNet batch 1, context = the prompt alone, no guidance combine."""
    g = golden("pipeline_txt2img_nocfg_2step")
    img, lat, final = _gen(models, dtype, steps=2, do_cfg=False)
    errs = [rel_l2(lat[i], g["latents"][i]) for i in range(2)]
    assert max(errs) < TOL[dtype], errs
    _check_final_image(models, img, final, g, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_denoise_steps_teacher_forced(models, dtype):
    """Each step fed the REFERENCE's previous latents (and the host noise stream): one UNet forward +
    CFG + DDPM step per comparison, no error carried between steps."""
    g = golden("pipeline_txt2img_3step")
    for name, m in models.items():
        with_dtype(m, FP32 if name == "clip" else dtype)      # as pipeline.generate does
    tok = synth.SyntheticTokenizer()
    pos = pipeline._token_tensor(tok, ["synthetic:0"], DEV)
    neg = pipeline._token_tensor(tok, [""], DEV)
    ctx = pipeline.encode_prompts(models["clip"], pos, neg)
    ts, coef, noise, lat0 = pipeline.prepare_latents([42], pipeline.DDPMSampler, 3, (1, 4, 64, 64), torch.device(DEV))
    loop = pipeline.DenoiseLoop(models["diffusion"], ctx, ts, coef, noise, lat0, 7.5)
    errs = []
    for i in range(3):
        start = lat0 if i == 0 else torch.from_numpy(g["latents"][i - 1]).to(DEV)
        loop.reset(start, step=i)
        loop.step()
        errs.append(rel_l2(loop.latents.cpu(), g["latents"][i]))
    print(f"\n[{dtype}] teacher-forced rel-L2: " + " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < TOL[dtype], errs


def test_generate_graph_equals_eager(models):
    """Three ways through the step loop, bit-equal: ONE hipGraph launch for the whole schedule (what generate() does
    without a trace / preview), one-step graph replays, and eager launches."""
    a, la, _ = _gen(models, FP32, steps=3, use_graph=True)         # (trace requested: one-step graph replays)
    b, lb, _ = _gen(models, FP32, steps=3, use_graph=False)
    assert torch.equal(la, lb) and np.array_equal(a, b)
    pipeline.clear_loop_cache()
    kw = dict(models=models, seed=42, device="cuda", sampler="ddpm", cfg_scale=7.5, steps=3, tokenizer=synth.SyntheticTokenizer(),
              compute_dtype=FP32, return_latents=True)
    c, lc = pipeline.generate("synthetic:0", "", **kw)
    loop = next(reversed(pipeline._LOOP_CACHE.values()))[1]
    assert loop.full_graph is not None and loop.host_step == loop.n_steps      # the whole loop went out as one launch
    assert torch.equal(lc.cpu(), la[-1]) and np.array_equal(c, a)
    c2, lc2 = pipeline.generate("synthetic:0", "", **kw)                       # replay of the cached whole-loop graph
    assert torch.equal(lc2, lc) and np.array_equal(c2, c)
    d, ld = pipeline.generate("synthetic:0", "", **{**kw, "cfg_scale": 3.0})   # another guidance scale: same captured graph,
    assert len(pipeline._LOOP_CACHE) == 1 and not torch.equal(ld, lc)          # the scale rides in the coefficient table
    pipeline.clear_loop_cache()
    e, le = pipeline.generate("synthetic:0", "", **{**kw, "cfg_scale": 3.0, "use_graph": False})
    assert torch.equal(ld, le) and np.array_equal(d, e)


class _f32_mode:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = ops.set_f32_mode(self.mode)
        pipeline.clear_loop_cache()

    def __exit__(self, *exc):
        ops.set_f32_mode(self.old)
        pipeline.clear_loop_cache()


def test_exact_f32_mode_teacher_forced_and_50step_free_running(models):
    """The `exact` fp32 mode (v_mfma_f32_16x16x4_f32; bench.py reports its step latency) through the whole pipeline:
    per-step teacher-forced and the 50-step free-running run of BASELINE config 1 against the reference, fp32 budget."""
    with _f32_mode("exact"):
        test_denoise_steps_teacher_forced(models, FP32)
        test_generate_txt2img_50step_vs_reference(models, FP32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_img2img_vs_reference(models, dtype):
    from PIL import Image
    g = golden("pipeline_img2img_5step_s0998")
    pil = Image.fromarray(np.random.default_rng(7).integers(0, 256, (512, 512, 3), dtype=np.uint8))
    img, lat, final = _gen(models, dtype, steps=5, input_image=pil, strength=0.998)
    assert lat.shape[0] == 3 == int(g["n_steps"])
    errs = [rel_l2(lat[i], g["latents"][i]) for i in range(3)]
    print(f"\n[{dtype}] img2img free-running rel-L2: " + " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < TOL[dtype], errs
    _check_final_image(models, img, final, g, dtype)
    with pytest.raises(IndexError, match="leaves no timesteps"):       # the reference's set_strength bug, kept
        _gen(models, dtype, steps=50, input_image=pil, strength=0.8)


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_txt2img_50step_vs_reference(models, dtype):
    """BASELINE config 1 vs the reference CPU run: 512^2, 50-step DDPM, CFG 7.5, seed 42, free-running."""
    g = golden("pipeline_txt2img_50step")
    img, lat, final = _gen(models, dtype, steps=50)
    errs = [rel_l2(lat[int(s)], g["latents"][j]) for j, s in enumerate(g["latent_steps"])]
    print(f"\n[{dtype}] 50-step free-running rel-L2 per recorded step: " + " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < TOL[dtype], errs
    _check_final_image(models, img, final, g, dtype)


def test_generate_batch_is_shard_invariant(models):
    """Extension: a batch is B independent replicas; image i uses seed+i and does not depend on
    which other images share its batch (the property multi-GPU sharding relies on).  The GEMM
    launcher picks tile / split-K by problem size, so the summation order (not the math) can change
    with the batch: equal to summation-order rounding (fp32 1e-5, bf16 its tolerance); the same
    batch shape is bit-reproducible (test_generate_graph_equals_eager)."""
    tok = synth.SyntheticTokenizer()
    for dtype in (FP32, BF16):
        kw = dict(models=models, device="cuda", steps=2, cfg_scale=7.5, tokenizer=tok, compute_dtype=dtype, return_latents=True)
        both, lat_both = pipeline.generate(["synthetic:0", "synthetic:1"], "", seed=42, **kw)
        one, lat_one = pipeline.generate(["synthetic:1"], "", seed=43, **kw)
        assert both.shape == (2, 512, 512, 3)
        if dtype == FP32:
            assert rel_l2(lat_both[1].cpu(), lat_one[0].cpu()) < 1e-5
            d = np.abs(both[1].astype(np.int32) - one[0].astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() < 0.01
        else:
            assert rel_l2(lat_both[1].cpu(), lat_one[0].cpu()) < TOL[BF16]
            assert rel_l2(both[1].astype(np.float32), one[0].astype(np.float32)) < TOL[BF16]


def test_generate_errors_match_reference(models):
    kw = dict(models=models, device="cuda", tokenizer=synth.SyntheticTokenizer())
    with pytest.raises(ValueError, match="Strength must be between 0 and 1"):
        pipeline.generate("synthetic:0", "", strength=0.0, **kw)
    with pytest.raises(ValueError, match="Sampler 'heun' not found"):      # pipeline.py:84
        pipeline.generate("synthetic:0", "", sampler="heun", steps=2, **kw)
    with pytest.raises(Exception, match="HIP device"):
        pipeline.generate("synthetic:0", "", models=models, device="cpu", tokenizer=synth.SyntheticTokenizer())


# ------------------------------------------------------------------ converter + loader
def test_loader_from_standard_checkpoint_end_to_end(models, tmp_path):
    """The hashed weights written out in CompVis checkpoint layout (full SD v1.5 size, .safetensors),
    read back through loader.preload_models_from_standard_weights: every parameter bit-equal, and the
    loaded models reproduce the reference's golden outputs."""
    from safetensors.torch import save_file
    from sd_amd import loader, model_converter
    std = model_converter.to_standard_weights({k: {n: t.cpu() for n, t in m.state_dict().items()} for k, m in models.items()})
    path = str(tmp_path / "v1-5-synthetic.safetensors")
    save_file({k: v.contiguous() for k, v in std.items()}, path)
    del std
    loaded = loader.preload_models_from_standard_weights(path, device=DEV)
    assert set(loaded) == {"clip", "encoder", "decoder", "diffusion"}
    for name, m in loaded.items():
        want = models[name].state_dict()
        got = m.state_dict()
        assert set(got) == set(want), name
        for k in want:
            assert got[k].is_cuda and torch.equal(got[k], want[k]), (name, k)
    g = golden("clip")
    ctx = loaded["clip"](torch.from_numpy(g["tokens"]).to(DEV))
    assert rel_l2(ctx.cpu(), g["out"]) < TOL[FP32]
    gd = golden("diffusion")
    out = loaded["diffusion"](hin("diff.lat.16", (2, 4, 16, 16), 2.0).to(DEV), ctx, pipeline.get_time_embedding(500).to(DEV))
    assert rel_l2(out.cpu(), gd["out16"]) < TOL[FP32]
    dec = loaded["decoder"](hin("dec.lat.8", (2, 4, 8, 8), 0.5).to(DEV))
    assert rel_l2(dec.cpu(), golden("vae_decoder")["dec8"]) < TOL[FP32]


# ------------------------------------------------------------------ BASELINE configs 4 and 5 at their own sizes
_ORACLE_CACHE = {}


def _cpu_sds(models):
    return cpu_state_dicts(models)


@pytest.mark.parametrize("dtype", DTYPES)
def test_diffusion_at_latent_96(models, dtype):
    """Config 5's UNet problem (768^2 image -> latent 96x96, 9216-token self-attention), batch 2, against the
    CPU oracle at full size."""
    lat, t320 = hin("diff.lat.96", (2, 4, 96, 96), 2.0), pipeline.get_time_embedding(740)
    ctx = torch.from_numpy(golden("clip")["out"])
    if "d96" not in _ORACLE_CACHE:
        _ORACLE_CACHE["d96"] = O.diffusion_forward(_cpu_sds(models)["diffusion"], lat, ctx, t320)
    out = with_dtype(models["diffusion"], dtype)(lat.to(DEV), ctx.to(DEV), t320.to(DEV))
    assert tuple(out.shape) == (2, 4, 96, 96)
    assert rel_l2(out.cpu(), _ORACLE_CACHE["d96"]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_768(models, dtype, monkeypatch):
    """Config 5 end to end at batch 1: the reference switches resolution through its module constants
    (pipeline.py:9-13); one CFG step + VAE decode to 768x768 against the oracle."""
    for name, v in (("WIDTH", 768), ("HEIGHT", 768), ("LATENTS_WIDTH", 96), ("LATENT_HEIGHT", 96)):
        monkeypatch.setattr(pipeline, name, v)
    if "g768" not in _ORACLE_CACHE:
        tr = []
        img = O.generate(_cpu_sds(models), synth.synthetic_token_ids(0), synth.EMPTY_PROMPT_IDS, seed=42, steps=1,
                         latent_hw=(96, 96), trace=tr)
        _ORACLE_CACHE["g768"] = (img, tr[-1])
    want_img, want_lat = _ORACLE_CACHE["g768"]
    img, lat, final = _gen(models, dtype, steps=1)
    assert img.shape == (768, 768, 3) and img.dtype == np.uint8
    assert rel_l2(lat[-1], want_lat) < TOL[dtype]
    diff = np.abs(img.astype(np.int32) - want_img.astype(np.int32))
    if dtype == FP32:
        assert diff.max() <= 1 and (diff > 0).mean() < 0.01
    else:
        assert rel_l2(img.astype(np.float32), want_img.astype(np.float32)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_768_ten_steps_free_running(models, dtype, monkeypatch):
    """Config 5's problem (768^2, latent 96x96) free-running over the first 10 steps of the 50-step schedule at batch 1,
    every step's latents against the CPU oracle (the one-step test above cannot see error carried between steps)."""
    for name, v in (("WIDTH", 768), ("HEIGHT", 768), ("LATENTS_WIDTH", 96), ("LATENT_HEIGHT", 96)):
        monkeypatch.setattr(pipeline, name, v)
    n = 10
    sds = _cpu_sds(models)
    if "g768x10" not in _ORACLE_CACHE:
        toks = torch.tensor([synth.synthetic_token_ids(0), synth.EMPTY_PROMPT_IDS])
        ctx = O.clip_forward(sds["clip"], toks)
        samp = O.DDPMSampler(torch.Generator().manual_seed(42))
        samp.set_steps(50)
        lat = torch.randn((1, 4, 96, 96), generator=samp.generator)
        tr = []
        for t in samp.timesteps[:n]:
            out = O.diffusion_forward(sds["diffusion"], lat.repeat(2, 1, 1, 1), ctx, O.get_time_embedding(int(t)))
            lat = samp.step(int(t), lat, 7.5 * (out[0:1] - out[1:2]) + out[1:2])
            tr.append(lat.clone())
        _ORACLE_CACHE["g768x10"] = torch.stack(tr)
    want = _ORACLE_CACHE["g768x10"]
    # the product side: generate()'s own pieces, stopped after n of the 50 steps
    for name, m in models.items():
        with_dtype(m, FP32 if name == "clip" else dtype)
    tok = synth.SyntheticTokenizer()
    ctx = pipeline.encode_prompts(models["clip"], pipeline._token_tensor(tok, ["synthetic:0"], DEV), pipeline._token_tensor(tok, [""], DEV))
    ts, coef, noise, lat0 = pipeline.prepare_latents([42], pipeline.DDPMSampler, 50, (1, 4, 96, 96), torch.device(DEV))
    loop = pipeline.DenoiseLoop(models["diffusion"], ctx, ts, coef, noise, lat0, 7.5)
    errs = []
    for i in range(n):
        loop.step()
        errs.append(rel_l2(loop.latents.cpu(), want[i]))
    print(f"\n[{dtype}] 768^2 free-running rel-L2, steps 1-{n}: " + " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < TOL[dtype], errs


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_img2img_strength08_batch(models, dtype):
    """Config 4's workload (img2img, strength 0.8) needs the intended strength semantics (the literal ones leave no
    timesteps, see test_generate_img2img_vs_reference): batch of 2 images with per-image seeds, 5-step schedule ->
    the last 4 steps, each image against an independent oracle run."""
    from PIL import Image
    rng = np.random.default_rng(11)
    pils = [Image.fromarray(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)) for _ in range(2)]
    for m in models.values():
        with_dtype(m, dtype)
    imgs, final = pipeline.generate(["synthetic:0", "synthetic:1"], "", input_image=pils, strength=0.8, models=models, seed=42,
                                    device="cuda", sampler="ddpm", steps=5, cfg_scale=7.5, tokenizer=synth.SyntheticTokenizer(),
                                    compute_dtype=dtype, strength_semantics="intended", return_latents=True)
    assert imgs.shape == (2, 512, 512, 3)
    for i in range(2):
        key = f"i2i{i}"
        if key not in _ORACLE_CACHE:
            tr = []
            img = O.generate(_cpu_sds(models), synth.synthetic_token_ids(i), synth.EMPTY_PROMPT_IDS, seed=42 + i, steps=5,
                             input_image=pils[i], strength_start=1, trace=tr)
            _ORACLE_CACHE[key] = (img, tr[-1])
        want_img, want_lat = _ORACLE_CACHE[key]
        assert rel_l2(final[i:i + 1].cpu(), want_lat) < TOL[dtype], i
        diff = np.abs(imgs[i].astype(np.int32) - want_img.astype(np.int32))
        if dtype == FP32:
            assert diff.max() <= 1 and (diff > 0).mean() < 0.01, i
        else:
            assert rel_l2(imgs[i].astype(np.float32), want_img.astype(np.float32)) < TOL[dtype], i


def test_generate_preview_callback(models):
    """The reference decodes + shows every step (pipeline.py:156-164); here that is the opt-in ``preview`` callback.
    It must not change the result, and its last frame is the returned image."""
    for m in models.values():
        with_dtype(m, FP32)
    kw = dict(models=models, seed=42, device="cuda", sampler="ddpm", steps=3, cfg_scale=7.5, tokenizer=synth.SyntheticTokenizer())
    frames = []
    img = pipeline.generate("synthetic:0", "", preview=lambda i, t, im: frames.append((i, t, im)), **kw)
    ref = pipeline.generate("synthetic:0", "", **kw)
    assert np.array_equal(img, ref)
    assert [f[0] for f in frames] == [0, 1, 2] and [f[1] for f in frames] == [666, 333, 0]
    assert frames[-1][2].shape == (1, 512, 512, 3) and np.array_equal(frames[-1][2][0], img)
    assert not np.array_equal(frames[0][2][0], img)
