"""Which GEMM-input roundings dominate the CFG-amplified bf16 error? (experiment, not product)"""
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
def run(t=666):
    out = m(lat2, ctx, pipeline.get_time_embedding(t).to(dev))
    return out, out[1:] + 7.5 * (out[:1] - out[1:])
rb = lambda x: x.bfloat16().float()
rh = lambda x: x.half().float()
g_gemm, g_conv, g_attn, g_geglu = ops.gemm, ops.conv3x3, ops.attention, ops.geglu
ref, refc = run()
def variant(name, gemm_sel=None, conv=False, rnd=rb, out_round=False):
    def gemm(a, w, bias=None, residual=None, *aa, **kw):
        N, K = w.shape
        big = a.numel() > 4096
        if big and gemm_sel is not None and gemm_sel(N, K, residual is not None): a = rnd(a)
        o = g_gemm(a, w, bias, residual, *aa, **kw)
        if out_round and big and residual is None: o = rnd(o)
        return o
    def conv3(x, w, bias, residual=None, *aa, **kw):
        o = g_conv(rnd(x) if conv else x, w, bias, residual, *aa, **kw)
        if out_round and residual is None: o = rnd(o)
        return o
    ops.gemm, ops.conv3x3 = gemm, conv3
    o, oc = run()
    ops.gemm, ops.conv3x3 = g_gemm, g_conv
    print(f"{name:44s} single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}", flush=True)
variant("none (sanity)")
variant("conv3x3 inputs only", conv=True)
variant("all gemm inputs only", gemm_sel=lambda N, K, r: True)
variant("gemm: geglu_1 (N=8C)", gemm_sel=lambda N, K, r: N == 8 * K)
variant("gemm: geglu_2 (K=4N)", gemm_sel=lambda N, K, r: K == 4 * N)
variant("gemm: qkv (N=3K)", gemm_sel=lambda N, K, r: N == 3 * K)
variant("gemm: square w/ residual (out_proj, conv_out)", gemm_sel=lambda N, K, r: N == K and r)
variant("gemm: square no residual (conv_in, q_proj)", gemm_sel=lambda N, K, r: N == K and not r)
variant("gemm: 1x1 skip (N!=K others)", gemm_sel=lambda N, K, r: N != K and N != 3 * K and N != 8 * K and K != 4 * N)
variant("all inputs bf16 (=E)", gemm_sel=lambda N, K, r: True, conv=True)
variant("all inputs bf16 + non-residual outputs bf16", gemm_sel=lambda N, K, r: True, conv=True, out_round=True)
variant("all inputs fp16", gemm_sel=lambda N, K, r: True, conv=True, rnd=rh)
variant("all inputs fp16 + non-res outputs fp16", gemm_sel=lambda N, K, r: True, conv=True, rnd=rh, out_round=True)
