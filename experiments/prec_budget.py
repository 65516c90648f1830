"""Error budget of the bf16 path (experiment, not product): where does the CFG-amplified error come from?
Emulates mixed-precision variants on the fp32 kernels by rounding operands to bf16 with torch."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import sd_amd
from sd_amd import clip, diffusion, engine, ops, pipeline, synth
torch.set_grad_enabled(False)
dev = "cuda"
rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
m = synth.load_hashed(diffusion.Diffusion(), "diffusion.").to(dev)
c = synth.load_hashed(clip.CLIP(), "clip.").to(dev)
tok = synth.SyntheticTokenizer()
pos = pipeline._token_tensor(tok, ["synthetic:0"], dev); neg = pipeline._token_tensor(tok, [""], dev)
ctx = pipeline.encode_prompts(c, pos, neg).float()
lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(42)).to(dev)
lat2 = torch.cat([lat, lat])
def run(t):
    out = m(lat2, ctx, pipeline.get_time_embedding(t).to(dev))
    return out, out[1:] + 7.5 * (out[:1] - out[1:])
orig = {k: v.clone() for k, v in m.state_dict().items()}
rb = lambda x: x.bfloat16().float()
g_gemm, g_conv, g_attn = ops.gemm, ops.conv3x3, ops.attention
for t in (666, 980, 20):
    engine.set_compute_dtype(m, torch.float32); engine.clear_cache(); m.load_state_dict(orig)
    ref, refc = run(t)
    print(f"t={t} |pos-neg|/|neg| = {rl2(ref[:1], ref[1:]):.3e}")
    engine.set_compute_dtype(m, torch.bfloat16)
    o, oc = run(t); print(f"  A pure bf16:            single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}")
    engine.set_compute_dtype(m, torch.float32)
    # B: weights rounded only
    for k, p in m.named_parameters():
        if p.dim() >= 2: p.data.copy_(rb(p.data))
    engine.clear_cache()
    o, oc = run(t); print(f"  B f32, bf16 weights:    single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}")
    # C: + GEMM/conv activation operands rounded (f32 trunk)
    ops.gemm = lambda a, w, *aa, **kw: g_gemm(rb(a) if a.shape[0] * 0 + a.numel() > 4096 else a, w, *aa, **kw)
    ops.conv3x3 = lambda x, *aa, **kw: g_conv(rb(x), *aa, **kw)
    o, oc = run(t); print(f"  C B + bf16 GEMM inputs: single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}")
    ops.attention = lambda q, k, v, *aa, **kw: g_attn(rb(q), rb(k), rb(v), *aa, **kw)
    o, oc = run(t); print(f"  D C + bf16 q,k,v:       single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}")
    ops.gemm, ops.conv3x3, ops.attention = g_gemm, g_conv, g_attn
    # E: f32 weights, only activations rounded at GEMM inputs
    m.load_state_dict(orig); engine.clear_cache()
    ops.gemm = lambda a, w, *aa, **kw: g_gemm(rb(a) if a.numel() > 4096 else a, w, *aa, **kw)
    ops.conv3x3 = lambda x, *aa, **kw: g_conv(rb(x), *aa, **kw)
    o, oc = run(t); print(f"  E f32 w, bf16 GEMM in:  single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}")
    ops.gemm, ops.conv3x3, ops.attention = g_gemm, g_conv, g_attn
