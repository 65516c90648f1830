"""BASELINE configs 4 and 5 on ONE GPU (the driver's bench line is config 3; these are for DESIGN.md).
  config 4: img2img, strength 0.8 of a 50-step schedule (intended semantics: the last 40 steps), VAE encode +
            add_noise + 40 x [UNet + CFG + DDPM] + VAE decode, per-GPU batches of the 64-image job at N = 8/4/1
  config 5: 768x768 txt2img (latent 96x96, 9216-token self-attention), 50 steps, per-GPU batches
usage: bench_configs.py [4|5] [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from sd_amd import encoder, pipeline, synth, engine
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batches = [int(b) for b in sys.argv[2:]] or ([8, 16, 64] if cfg == 4 else [8, 32])
models, _ = bench.build_models(dev, torch.bfloat16, False)
if cfg == 4:
    models["encoder"] = engine.set_compute_dtype(synth.load_hashed(encoder.VAE_Encoder(), "encoder.").to(dev), torch.bfloat16)
hw = 64 if cfg == 4 else 96
gflop_img = {4: 67918.0, 5: 220591.0}[cfg]
for B in batches:
    ids = torch.tensor([synth.synthetic_token_ids(i) for i in range(B)], dtype=torch.long, device=dev)
    neg = torch.tensor([synth.EMPTY_PROMPT_IDS] * B, dtype=torch.long, device=dev)
    seeds = [42 + i for i in range(B)]
    imgs = None
    if cfg == 4:
        rng = np.random.default_rng(7)
        imgs = torch.from_numpy(rng.integers(0, 256, (B, 512, 512, 3), dtype=np.uint8)).to(dev)

    def prepare():
        return pipeline.prepare_latents(seeds, pipeline.DDPMSampler, 50, (1, 4, hw, hw), dev, models.get("encoder"), imgs, 0.8, "intended")

    ts, coef, noise, lat = prepare()
    ctx = pipeline.encode_prompts(models["clip"], ids, neg)
    loop = pipeline.DenoiseLoop(models["diffusion"], ctx, ts, coef, noise, lat, 7.5, True, True)

    def job():
        # everything of generate() that touches the GPU per request: (encode + add_noise,) CLIP, the steps, decode
        if cfg == 4:
            _, _, _, lat0 = prepare()
        else:
            lat0 = lat
        loop.set_context(pipeline.encode_prompts(models["clip"], ids, neg))
        loop.reset(lat0)
        final = loop.run()
        return pipeline.decode_to_uint8(models["decoder"], final)

    job(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = job(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    step_ms = bench.unet_step_ms(loop, lat)
    print(f"config {cfg} batch {B}: {B / dt:6.2f} images/s  ({len(ts)} steps, UNet step {step_ms:.1f} ms at UNet batch {2 * B}, "
          f"{B / dt * gflop_img / 1e3:.0f} TFLOP/s effective, out {tuple(out.shape)})", flush=True)
    del loop
    torch.cuda.empty_cache()
