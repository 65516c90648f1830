"""Median time of one captured UNet step (+CFG/DDPM) at UNet batch 2*B, for A/B runs of two library builds
(SDAMD_LIB=<other libsdamd.so> python experiments/step_time.py [B] [f32]); prints one line."""
import sys; sys.path.insert(0, '.')
import os, torch, sd_amd
from sd_amd import clip, diffusion, engine, ops, pipeline, synth
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
dev = torch.device("cuda")
dm = engine.set_compute_dtype(synth.load_hashed(diffusion.Diffusion(), "diffusion.").to(dev), DT)
cm = synth.load_hashed(clip.CLIP(), "clip.").to(dev)
pos = torch.tensor([synth.synthetic_token_ids(i) for i in range(B)], device=dev)
neg = torch.tensor([synth.EMPTY_PROMPT_IDS] * B, device=dev)
ctx = pipeline.encode_prompts(cm, pos, neg)
ts, coef, noise, lat = pipeline.prepare_latents([42 + i for i in range(B)], pipeline.DDPMSampler, 50, (1, 4, 64, 64), dev)
loop = pipeline.DenoiseLoop(dm, ctx, ts, coef, noise, lat, 7.5)
res = []
for r in range(7):
    loop.reset(lat); loop.step(); loop.step()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(12): loop.step()
    e.record(); torch.cuda.synchronize()
    res.append(s.elapsed_time(e) / 12)
res.sort()
fp = float(loop.latents.double().abs().sum())
print(f"{os.environ.get('SDAMD_LIB', 'in-tree')}: UNet step median {res[3]:.3f} ms min {res[0]:.3f} (batch {2*B}, {DT}) latents |sum| {fp:.6f}")
