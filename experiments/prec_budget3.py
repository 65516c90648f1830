"""Emulate the planned mixed scheme: f32 trunk, bf16 MFMA operands, hi+lo split on trunk-through-linear GEMMs."""
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
ref, refc = run()
orig = {k: v.clone() for k, v in m.state_dict().items()}
e_lin, e_c1, e_c3, o_attn, o_geglu = engine.linear, engine.conv1x1, engine.conv3x3, ops.attention, ops.geglu
def scheme(name, split_trunk, round_h, wround=True, split_outproj=False, split_geglu2=False):
    m.load_state_dict(orig)
    if wround:
        for k, p in m.named_parameters():
            if p.dim() >= 2: p.data.copy_(rb(p.data))
    engine.clear_cache()
    def linear(x, lin, residual=None, *a, **kw):
        keep = (split_outproj and residual is not None and lin.in_features == lin.out_features) or \
               (split_geglu2 and lin.in_features == 4 * lin.out_features)
        o = e_lin(x if keep else rb(x), lin, residual, *a, **kw)
        return o if residual is not None else rb(o)
    def conv1x1(x, conv, residual=None):
        trunk_in = residual is not None or conv.in_channels != conv.out_channels
        return e_c1(x if (trunk_in and split_trunk) else rb(x), conv, residual)
    def conv3x3(x, conv, residual=None, stride=1, pad_mode=0, upsample=False, bias=None):
        o = e_c3(rb(x), conv, residual, stride, pad_mode, upsample, bias)
        return rb(o) if (round_h and bias is not None and residual is None) else o
    engine.linear, engine.conv1x1, engine.conv3x3 = linear, conv1x1, conv3x3
    ops.attention = lambda q, k, v, *a, **kw: rb(o_attn(q, k, v, *a, **kw))
    ops.geglu = lambda x: rb(o_geglu(x))
    o, oc = run()
    engine.linear, engine.conv1x1, engine.conv3x3, ops.attention, ops.geglu = e_lin, e_c1, e_c3, o_attn, o_geglu
    print(f"{name:60s} single {rl2(o, ref):.2e}  cfg {rl2(oc, refc):.2e}", flush=True)
scheme("f32 trunk, no split, h f32", False, False)
scheme("f32 trunk, no split, h bf16", False, True)
scheme("f32 trunk, split skip+conv_output, h f32", True, False)
scheme("f32 trunk, split skip+conv_output, h bf16", True, True)
scheme("f32 trunk, split skip+conv_output+out_proj, h f32", True, False, split_outproj=True)
scheme("f32 trunk, split skip+conv_output+out_proj+geglu2, h f32", True, False, split_outproj=True, split_geglu2=True)
scheme("same as 3 but f32 weights", True, False, wround=False)
