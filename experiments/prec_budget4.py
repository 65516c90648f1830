"""Error budget of the real f32-trunk/bf16-operand implementation, by module class (experiment)."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import sd_amd
from sd_amd import clip, diffusion, engine, ops, pipeline, synth
torch.set_grad_enabled(False)
dev = "cuda"
F32, BF16 = torch.float32, torch.bfloat16
rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
m = synth.load_hashed(diffusion.Diffusion(), "diffusion.").to(dev)
c = synth.load_hashed(clip.CLIP(), "clip.").to(dev)
tok = synth.SyntheticTokenizer()
pos = pipeline._token_tensor(tok, ["synthetic:0"], dev); neg = pipeline._token_tensor(tok, [""], dev)
lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(42)).to(dev)
lat2 = torch.cat([lat, lat])
def run(ctx, t=666):
    out = m(lat2, ctx, pipeline.get_time_embedding(t).to(dev))
    return out, out[1:] + 7.5 * (out[:1] - out[1:])
engine.set_compute_dtype(c, F32); ctx32 = pipeline.encode_prompts(c, pos, neg)
engine.set_compute_dtype(c, BF16); ctx16 = pipeline.encode_prompts(c, pos, neg)
print("clip bf16 ctx err", rl2(ctx16.float(), ctx32))
engine.set_compute_dtype(m, F32); ref, refc = run(ctx32)
def setcls(classes, dt):
    for mod in m.modules():
        if isinstance(mod, classes):
            for sub in mod.modules():
                if isinstance(sub, engine.HipModule): sub.compute_dtype = dt
def report(name, ctx):
    o, oc = run(ctx); print(f"{name:50s} single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}", flush=True)
engine.set_compute_dtype(m, BF16); report("all bf16, bf16 clip", ctx16)
report("all bf16, f32 clip", ctx32)
engine.set_compute_dtype(m, BF16); setcls((diffusion.UNET_AttentionBlock,), F32); report("attn blocks f32 (f32 clip)", ctx32)
engine.set_compute_dtype(m, BF16); setcls((diffusion.UNET_ResidualBlock,), F32); report("res blocks f32 (f32 clip)", ctx32)
engine.set_compute_dtype(m, BF16); setcls((diffusion.UNET_ResidualBlock, diffusion.UNET_AttentionBlock), F32); report("res+attn blocks f32: up/down/final bf16", ctx32)
engine.set_compute_dtype(m, F32); report("all f32 with bf16-clip ctx", ctx16)
# inside attention blocks: which part? patch engine pieces to f32 selectively is harder; use per-stage split instead
for lo, hi, nm in ((0, 4, "enc 64x64"), (4, 7, "enc 32x32"), (7, 12, "enc 16/8")):
    engine.set_compute_dtype(m, F32)
    for i in range(lo, hi):
        for sub in m.unet.encoders[i].modules():
            if isinstance(sub, engine.HipModule): sub.compute_dtype = BF16
    report(f"only {nm} bf16", ctx32)
engine.set_compute_dtype(m, F32)
for sub in m.unet.bottleneck.modules():
    if isinstance(sub, engine.HipModule): sub.compute_dtype = BF16
report("only bottleneck bf16", ctx32)
for lo, hi, nm in ((0, 3, "dec 8"), (3, 6, "dec 16"), (6, 9, "dec 32"), (9, 12, "dec 64")):
    engine.set_compute_dtype(m, F32)
    for i in range(lo, hi):
        for sub in m.unet.decoders[i].modules():
            if isinstance(sub, engine.HipModule): sub.compute_dtype = BF16
    report(f"only {nm} bf16", ctx32)
