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
    """Base of every module that executes through libsdamd.  ``compute_dtype`` is the type of the
    MFMA operands (weights and normalised activations) on the device: float32 or bfloat16.

    ``f32_trunk`` (class attribute): when the compute dtype is bfloat16, keep the residual stream
    ("trunk": every tensor that a residual add accumulates into) in float32.  GroupNorm /
    LayerNorm read the f32 trunk and emit bf16 operands, GEMM/conv epilogues add the f32 residual
    and write f32.  The UNet and CLIP need it: classifier-free guidance amplifies the UNet's
    rounding noise ~10x (measured: pure bf16 3.6e-2 on eps, this scheme 1.7e-2; experiments/)."""

    compute_dtype = torch.float32
    f32_trunk = False

    @property
    def trunk_dtype(self):
        return torch.float32 if self.f32_trunk else self.compute_dtype


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
# Dtype rule of every GEMM-like helper: the operand ``x`` has the compute dtype; the result has
# ``out_dtype`` (default: the residual's dtype if there is one, else x's); a residual has the
# result's dtype.
def _out_f32(x, residual, out_dtype):
    if out_dtype is None:
        out_dtype = x.dtype if residual is None else residual.dtype
    return out_dtype == torch.float32 and x.dtype != torch.float32


def linear(x, lin, residual=None, act=ops.ACT_NONE, out_dtype=None, out=None):
    return ops.gemm(x, linear_w(lin, x.dtype), bias_of(lin), residual, act, _out_f32(x, residual, out_dtype), out=out)


def linear_ln(x, lin, residual, ln, out=None, ln_out=None):
    """(t, n) = (linear(x) + residual in f32, LayerNorm(t) as the bf16 operand of the next projection): one launch where
    ops.gemm_ln_fused says so, else the GEMM followed by the LayerNorm kernel (``ln_out`` only with the fused form)."""
    M = x.numel() // x.shape[-1]
    if x.dtype == torch.bfloat16 and ops.gemm_ln_fused(M, lin.out_features, x.dtype):
        return ops.gemm_ln(x, linear_w(lin, x.dtype), bias_of(lin), residual, f32(ln.weight), f32(ln.bias), ln.eps, out=out, ln_out=ln_out)
    assert ln_out is None
    t = ops.gemm(x, linear_w(lin, x.dtype), bias_of(lin), residual, ops.ACT_NONE, _out_f32(x, residual, torch.float32), out=out)
    return t, layernorm(t, ln, x.dtype)


def conv1x1_ln(x, conv, ln):
    """conv1x1 (bf16 operand -> f32 trunk) + LayerNorm of the result, as linear_ln."""
    M = x.numel() // x.shape[-1]
    if x.dtype == torch.bfloat16 and ops.gemm_ln_fused(M, conv.out_channels, x.dtype):
        return ops.gemm_ln(x, conv1x1_w(conv, x.dtype), bias_of(conv), None, f32(ln.weight), f32(ln.bias), ln.eps)
    t = conv1x1(x, conv, out_dtype=torch.float32)
    return t, layernorm(t, ln, x.dtype)


def linear_geglu(x, lin):
    """diffusion.py:127-130: x, gate = linear_geglu_1(x).chunk(2, -1); x * gelu(gate) -- as one GEMM
    whose epilogue gates the f32 accumulators when the kernel family supports it (bf16), else the
    GEMM followed by the GeGLU kernel."""
    if ops.gemm_supports_act(ops.ACT_GEGLU, x.dtype, lin.out_features):
        w = packed(lin.weight, "geglu_w", x.dtype, lambda t: ops.geglu_pack(t.to(x.dtype)))
        b = packed(lin.bias, "geglu_b", torch.float32, lambda t: ops.geglu_pack(t.float()))
        return ops.gemm(x, w, b, None, ops.ACT_GEGLU)
    return ops.geglu(linear(x, lin))


def linear_hilo(x, lin, residual):
    """linear(x) + residual (f32) emitted directly as the bf16 hi|lo pair [.., 2N] that conv1x1_presplit takes, or
    None when this problem does not run on the kernels that can (the caller then uses linear + conv1x1)."""
    M = x.numel() // x.shape[-1]
    if x.dtype != torch.bfloat16 or residual is None or residual.dtype != torch.float32 or \
            not ops.gemm_hilo_fused(M, lin.out_features, lin.in_features, x.dtype):
        return None
    return ops.gemm(x, linear_w(lin, x.dtype), bias_of(lin), residual, ops.ACT_NONE, out_hilo=True)


def conv1x1_presplit(xs, conv, residual, want_stats=False, out=None):
    """conv1x1 of an operand that already is the hi|lo pair [..., 2C] (see conv1x1): f32 residual, f32 result."""
    w2 = packed(conv.weight, "c1x2", torch.bfloat16,
                lambda w: w.reshape(w.shape[0], w.shape[1]).to(torch.bfloat16).repeat(1, 2).contiguous())
    return ops.gemm(xs, w2, bias_of(conv), residual, ops.ACT_NONE, True, flops_k=conv.in_channels, want_stats=want_stats, out=out)


def conv1x1(x, conv, residual=None, out_dtype=None, compute_dtype=None, x2=None, want_stats=False, out=None):
    """x NHWC [B,H,W,C].  An f32 ``x`` under a bf16 compute dtype is the residual trunk itself:
    it goes through the MFMA as two bf16 terms (hi|lo) against [W|W] instead of being rounded.
    ``x2`` (that mode only): the operand is the channel concat [x | x2]."""
    if compute_dtype == torch.bfloat16 and x.dtype == torch.float32:
        w2 = packed(conv.weight, "c1x2", torch.bfloat16,
                    lambda w: w.reshape(w.shape[0], w.shape[1]).to(torch.bfloat16).repeat(1, 2).contiguous())
        xs = ops.split_hilo(x, x2)
        return ops.gemm(xs, w2, bias_of(conv), residual, ops.ACT_NONE, _out_f32(xs, residual, out_dtype), flops_k=conv.in_channels,
                        want_stats=want_stats, out=out)
    assert x2 is None
    return ops.gemm(x, conv1x1_w(conv, x.dtype), bias_of(conv), residual, ops.ACT_NONE, _out_f32(x, residual, out_dtype),
                    want_stats=want_stats, out=out)


def conv3x3(x, conv, residual=None, stride=1, pad_mode=0, upsample=False, bias=None, out_dtype=None, want_stats=False):
    if upsample and residual is None and stride == 1 and pad_mode == 0 and \
            ops.gemm_supports_act(ops.ACT_GEGLU, torch.bfloat16, 256) and \
            (conv.out_channels % 160 == 0 or conv.out_channels % 256 == 0):
        # nearest-2x + conv3x3 as four 2x2 phase convs (4/9 of the MACs), see conv3x3_up_phase_w
        return ops.conv3x3(x, conv3x3_up_phase_w(conv, x.dtype, 1), bias_of(conv) if bias is None else bias, None, 1, 0, 2,
                           out_f32=_out_f32(x, residual, out_dtype))
    return ops.conv3x3(x, conv3x3_w(conv, x.dtype), bias_of(conv) if bias is None else bias, residual, stride,
                       pad_mode, upsample, out_f32=_out_f32(x, residual, out_dtype), want_stats=want_stats)


def conv3x3_up_phase_w(conv, dtype, dup):
    """OIHW -> [4 phases (py,px)][O][2][2][dup*I]: nearest-2x upsample + conv3x3(pad 1) restated as
    one 2x2 conv per output phase.  Output row 2y+py reads input rows {y-1: k0, y: k1+k2} (py=0) or
    {y: k0+k1, y+1: k2} (py=1); same along x.  The tap sums are formed in f32, then rounded."""
    def build(w):
        w = w.float()
        rows = (torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], 2), torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], 2))
        out = []
        for py in range(2):
            r = rows[py]                                           # [O, I, 2, 3]
            cols = (torch.stack([r[..., 0], r[..., 1] + r[..., 2]], 3), torch.stack([r[..., 0] + r[..., 1], r[..., 2]], 3))
            for px in range(2):
                k = cols[px].permute(0, 2, 3, 1)                   # [O, 2, 2, I]
                out.append(k.repeat(1, 1, 1, dup).reshape(k.shape[0], -1))
        return torch.stack(out).to(dtype).contiguous()
    return packed(conv.weight, f"c3up{dup}", dtype, build)


def conv3x3_trunk(x, conv, compute_dtype, stride=1, upsample=False):
    """conv3x3 whose operand is the residual trunk itself (UNet down/upsampling convs): result in
    x's dtype; an f32 trunk under bf16 compute goes through the MFMA as [hi|lo] channels against
    [W|W] per tap (see conv1x1)."""
    if compute_dtype == torch.bfloat16 and x.dtype == torch.float32:
        if upsample and ops.gemm_supports_act(ops.ACT_GEGLU, torch.bfloat16, 256) and \
                (conv.out_channels % 160 == 0 or conv.out_channels % 256 == 0):   # an 8-wave tile divides Cout
            return ops.conv3x3(ops.split_hilo(x), conv3x3_up_phase_w(conv, torch.bfloat16, 2), bias_of(conv), None, 1, 0, 2,
                               out_f32=True, flops_cin=conv.in_channels)
        w2 = packed(conv.weight, "c3x2", torch.bfloat16,
                    lambda w: w.permute(0, 2, 3, 1).to(torch.bfloat16).repeat(1, 1, 1, 2).reshape(w.shape[0], -1).contiguous())
        return ops.conv3x3(ops.split_hilo(x), w2, bias_of(conv), None, stride, 0, upsample, out_f32=True, flops_cin=conv.in_channels,
                           want_stats=True)
    if upsample and stride == 1 and ops.gemm_supports_act(ops.ACT_GEGLU, torch.bfloat16, 256) and \
            (conv.out_channels % 160 == 0 or conv.out_channels % 256 == 0) and ops._plan_rows(x.shape[0] * x.shape[1] * x.shape[2]) >= 256:
        # phase form, as conv3x3 above (below 256 low-resolution rows the f32 planner finds no split that fills the
        # chip and the gathered form on the 4-wave kernels is faster: 0.18 vs 0.81 ms at UNet batch 2, 8x8 -> 16x16)
        return ops.conv3x3(x, conv3x3_up_phase_w(conv, x.dtype, 1), bias_of(conv), None, 1, 0, 2)
    return ops.conv3x3(x, conv3x3_w(conv, x.dtype), bias_of(conv), None, stride, 0, upsample, want_stats=True)


def groupnorm(x, gn, silu, out_dtype=None, x2=None):
    return ops.groupnorm(x, f32(gn.weight), f32(gn.bias), gn.eps, silu, out_dtype, x2)


def layernorm(x, ln, out_dtype=None):
    return ops.layernorm(x, f32(ln.weight), f32(ln.bias), ln.eps, out_dtype)


def self_attention(x, attn, causal=False, residual=None, then_ln=None):
    """attention.py:25-75 on token-major x [B,S,C]: fused QKV GEMM -> flash attention -> out-proj GEMM
    (+ residual, whose dtype the result takes).  ``then_ln``: also return the LayerNorm of the result (linear_ln)."""
    qkv = linear(x, attn.in_proj)
    C = x.shape[-1]
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    o = ops.attention(q, k, v, attn.n_heads, causal)   # one flash kernel family, the VAE's 512-wide head included
    if then_ln is not None:
        return linear_ln(o, attn.out_proj, residual, then_ln)
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


def cross_attention(x, attn, kv, residual=None, then_ln=None):
    """attention.py:90-133 with precomputed ``kv`` = cross_kv(context, attn).  ``then_ln`` as for self_attention."""
    C = x.shape[-1]
    q = linear(x, attn.q_proj)
    o = ops.attention(q, kv[..., :C], kv[..., C:], attn.n_heads, False)
    if then_ln is not None:
        return linear_ln(o, attn.out_proj, residual, then_ln)
    return linear(o, attn.out_proj, residual)
