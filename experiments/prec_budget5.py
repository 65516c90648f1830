"""Error budget of the real implementation after the trunk fixes; attention-block internals; VAE decoder."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import sd_amd
from sd_amd import clip, decoder, diffusion, engine, ops, pipeline, synth
from conftest import golden
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
engine.set_compute_dtype(m, F32); ref, refc = run(ctx32)
def report(name):
    o, oc = run(ctx32); print(f"{name:56s} single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}", flush=True)
def setcls(classes, dt):
    for mod in m.modules():
        if isinstance(mod, classes):
            for sub in mod.modules():
                if isinstance(sub, engine.HipModule): sub.compute_dtype = dt
engine.set_compute_dtype(m, BF16); report("all bf16 (f32 clip)")
engine.set_compute_dtype(m, BF16); setcls((diffusion.UNET_AttentionBlock,), F32); report("attn blocks f32")
engine.set_compute_dtype(m, BF16); setcls((diffusion.UNET_ResidualBlock,), F32); report("res blocks f32")
engine.set_compute_dtype(m, BF16); setcls((diffusion.UNET_ResidualBlock, diffusion.UNET_AttentionBlock), F32); report("res+attn f32 (up/down/final bf16 split)")
# attention-block internals: run a sub-path in f32 by casting around it
engine.set_compute_dtype(m, BF16)
e_lin, e_sa, e_ca, e_c1 = engine.linear, engine.self_attention, engine.cross_attention, engine.conv1x1
def f32_linear(x, lin, residual=None, act=ops.ACT_NONE, out_dtype=None):
    o = e_lin(x.float(), lin, residual.float() if residual is not None else None, act)
    want = out_dtype or (residual.dtype if residual is not None else x.dtype)
    return ops.cast(o, want)
def geglu_f32(x, lin, residual=None, act=ops.ACT_NONE, out_dtype=None):
    if lin.out_features == 8 * lin.in_features:
        return e_lin(x.float(), lin)                 # f32 h -> geglu f32 -> geglu_2 f32
    if lin.in_features == 4 * lin.out_features:
        return e_lin(x, lin, residual, act, out_dtype)
    return e_lin(x, lin, residual, act, out_dtype)
engine.linear = geglu_f32; report("attn: geglu_1+gate+geglu_2 in f32"); engine.linear = e_lin
def sa32(x, attn, causal=False, residual=None):
    return e_sa(x.float(), attn, causal, residual)
engine.self_attention = sa32; report("attn: self-attention (qkv, attn, out_proj) in f32"); engine.self_attention = e_sa
def ca32(x, attn, kv, residual=None):
    return e_ca(x.float(), attn, kv.float(), residual)
engine.cross_attention = ca32; report("attn: cross-attention in f32"); engine.cross_attention = e_ca
def c1_32(x, conv, residual=None, out_dtype=None, compute_dtype=None):
    if x.dtype == BF16:   # conv_input (operand = GN output)
        return e_c1(x.float(), conv, residual, out_dtype, None)
    return e_c1(x, conv, residual, out_dtype, compute_dtype)
engine.conv1x1 = c1_32; report("attn: conv_input in f32"); engine.conv1x1 = e_c1
# VAE decoder
d = synth.load_hashed(decoder.VAE_Decoder(), "decoder.").to(dev)
g = golden("pipeline_txt2img_3step")
fl = torch.from_numpy(g["latents"][-1]).to(dev)
engine.set_compute_dtype(d, F32); dref = d(fl)
engine.set_compute_dtype(d, BF16); print("decoder bf16 rel-l2 on final latents:", rl2(d(fl), dref))
print("decoder f32 sensitivity: 1e-2 latent noise ->", rl2(engine.set_compute_dtype(d, F32)(fl * (1 + 0.01 * torch.randn_like(fl))), dref))
