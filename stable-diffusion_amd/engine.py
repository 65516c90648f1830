"""Weight repacking and shared building blocks of the HIP execution path.

The nn.Module classes of this package keep the reference's parameter names and PyTorch-native
layouts (the weights ABI, loader.py:18-27).  The kernels want other layouts (conv weights
[Cout][ky][kx][Cin], fused K|V projections, bf16 copies ...), so every parameter is repacked
once per (parameter version, device, compute dtype) and cached here.
"""
import weakref

import torch
import torch.nn as nn

from . import ops

_cache = {}


def packed(params, kind, dtype, build):
    """Cached repack of one or more parameters.  ``build(*tensors)`` runs once per version."""
    if not isinstance(params, (tuple, list)):
        params = (params,)
    key = (tuple(id(p) for p in params), kind, dtype)
    stamp = tuple((p._version, p.data_ptr()) for p in params)
    ent = _cache.get(key)
    if ent is not None and ent[1] == stamp and all(r() is p for r, p in zip(ent[0], params)):
        return ent[2]
    for p in params:
        if not p.is_cuda:
            raise ops._lib.SdamdError("module parameters must live on the HIP device: call .to('cuda') first")
    with torch.no_grad():
        value = build(*[p.detach() for p in params])
    _cache[key] = (tuple(weakref.ref(p) for p in params), stamp, value)
    return value


def clear_cache():
    _cache.clear()


class HipModule(nn.Module):
    """Base of every module that executes through libsdamd.  ``compute_dtype`` is the storage
    type of activations and weights on the device (float32 or bfloat16)."""

    compute_dtype = torch.float32


def set_compute_dtype(module, dtype):
    assert dtype in (torch.float32, torch.bfloat16)
    for m in module.modules():
        if isinstance(m, HipModule):
            m.compute_dtype = dtype
    return module


# ---- parameter views -------------------------------------------------------------------------
def f32(p):
    return packed(p, "f32", torch.float32, lambda t: t.float().contiguous())


def linear_w(lin, dtype):
    return packed(lin.weight, "lin", dtype, lambda w: w.to(dtype).contiguous())


def bias_of(mod):
    return None if mod.bias is None else f32(mod.bias)


def conv1x1_w(conv, dtype):
    return packed(conv.weight, "c1", dtype, lambda w: w.reshape(w.shape[0], w.shape[1]).to(dtype).contiguous())


def conv3x3_w(conv, dtype):
    """OIHW -> [O][ky][kx][I] flattened to [O, 9*I]."""
    return packed(conv.weight, "c3", dtype,
                  lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dtype).contiguous())


def conv3x3_cin_small_w(conv):
    """OIHW -> [ky][kx][I][O] f32."""
    return packed(conv.weight, "c3in", torch.float32, lambda w: w.permute(2, 3, 1, 0).float().contiguous())


def conv3x3_cout_small_w(conv, dtype):
    """OIHW -> [tap][I/V][O][V] with V = 16 bytes of dtype."""
    v = 8 if dtype == torch.bfloat16 else 4

    def build(w):
        o, i = w.shape[0], w.shape[1]
        t = w.permute(2, 3, 1, 0).reshape(9, i // v, v, o).permute(0, 1, 3, 2)
        return t.to(dtype).contiguous()

    return packed(conv.weight, "c3out", dtype, build)


# ---- shared blocks ---------------------------------------------------------------------------
def linear(x, lin, residual=None, act=ops.ACT_NONE, out_f32=False):
    return ops.gemm(x, linear_w(lin, x.dtype), bias_of(lin), residual, act, out_f32)


def conv1x1(x, conv, residual=None):
    """x NHWC [B,H,W,C]."""
    return ops.gemm(x, conv1x1_w(conv, x.dtype), bias_of(conv), residual)


def conv3x3(x, conv, residual=None, stride=1, pad_mode=0, upsample=False, bias=None):
    return ops.conv3x3(x, conv3x3_w(conv, x.dtype), bias_of(conv) if bias is None else bias, residual, stride,
                       pad_mode, upsample)


def groupnorm(x, gn, silu):
    return ops.groupnorm(x, f32(gn.weight), f32(gn.bias), gn.eps, silu)


def layernorm(x, ln):
    return ops.layernorm(x, f32(ln.weight), f32(ln.bias), ln.eps)


def self_attention(x, attn, causal=False, residual=None):
    """attention.py:25-75 on token-major x [B,S,C]: fused QKV GEMM -> flash attention (or the
    GEMM/softmax/GEMM form for the 512-wide single VAE head) -> out-proj GEMM (+ residual)."""
    qkv = linear(x, attn.in_proj)
    C = x.shape[-1]
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    if attn.d_head > 160:
        assert attn.n_heads == 1 and not causal
        o = ops.attention_unfused(q, k, v)
    else:
        o = ops.attention(q, k, v, attn.n_heads, causal)
    return linear(o, attn.out_proj, residual)


def cross_kv(ctx, attn):
    """K|V projections of the prompt context [B,77,768] as one GEMM -> [B,77,2C]; constant over
    the denoising loop, so pipeline.generate computes it once per prompt."""
    dtype = ctx.dtype
    w = packed((attn.k_proj.weight, attn.v_proj.weight), "kv", dtype,
               lambda k, v: torch.cat([k, v], 0).to(dtype).contiguous())
    b = None
    if attn.k_proj.bias is not None:
        b = packed((attn.k_proj.bias, attn.v_proj.bias), "kvb", torch.float32, lambda k, v: torch.cat([k, v]).float())
    return ops.gemm(ctx, w, b)


def cross_attention(x, attn, kv, residual=None):
    """attention.py:90-133 with precomputed ``kv`` = cross_kv(context, attn)."""
    C = x.shape[-1]
    q = linear(x, attn.q_proj)
    o = ops.attention(q, kv[..., :C], kv[..., C:], attn.n_heads, False)
    return linear(o, attn.out_proj, residual)
