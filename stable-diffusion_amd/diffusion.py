"""SD-v1.5 UNet behind the reference's ``Diffusion`` / ``UNET`` interfaces (diffusion.py:5-324).

Same class names, constructor arguments, forward signatures and state_dict keys as the
reference (so ``load_state_dict(strict=True)`` of its converter output works, loader.py:18-27),
but ``forward`` runs a flat sequence of libsdamd launches on NHWC activations:

  residual block  = GN+SiLU -> conv3x3(+bias+time bias) -> GN+SiLU -> conv3x3(+bias+skip)
  attention block = GN -> 1x1 GEMM -> [LN -> QKV GEMM -> flash attn -> out GEMM(+res)]
                    -> [LN -> Q GEMM -> flash cross attn -> out GEMM(+res)]
                    -> [LN -> GEMM -> GeGLU -> GEMM(+res)] -> 1x1 GEMM(+long res)

Channel concat of the skip connections, nearest-2x upsampling and all residual / bias adds ride
inside those kernels.  The per-block time projections ``linear_time(silu(t))`` are one GEMM over
the concatenated projection weights; the prompt K|V projections are computed once per prompt.
"""
import torch
import torch.nn as nn

from . import engine, ops
from .attention import CrossAttention, SelfAttention

_ATTENTION_TYPES = ("xformers", "flashattention", "default")  # all map to the one HIP implementation
# Classifier-free guidance runs the UNet on [latents; latents] with two prompts: until the first cross attention the two
# halves are the same tensor.  True: UNET.run(copies=2) computes that shared prefix once (A/B switch for experiments).
SHARE_CFG_PREFIX = True


class UNET_ResidualBlock(engine.HipModule):
    f32_trunk = True

    def __init__(self, input_channel: int, output_channel: int, n_time=1280):
        super().__init__()
        self.groupnorm_feature = nn.GroupNorm(32, input_channel)
        self.conv_feature = nn.Conv2d(input_channel, output_channel, kernel_size=3, padding=1)
        self.linear_time = nn.Linear(n_time, output_channel)
        self.groupnorm_merged = nn.GroupNorm(32, output_channel)
        self.conv_merged = nn.Conv2d(output_channel, output_channel, kernel_size=3, padding=1)
        self.residual_layer = (nn.Identity() if input_channel == output_channel
                               else nn.Conv2d(input_channel, output_channel, kernel_size=1, padding=0))

    def time_bias(self, time_silu):
        """conv_feature.bias + linear_time(silu(t)) as one f32 vector [Cout] (diffusion.py:30-36)."""
        w = engine.packed(self.linear_time.weight, "lin", torch.float32, lambda t: t.float().contiguous())
        b = engine.packed((self.linear_time.bias, self.conv_feature.bias), "tb", torch.float32, lambda a, c: (a + c).float())
        return ops.gemm(time_silu, w, b).reshape(-1)

    def run(self, x, tbias):
        """x NHWC in the trunk dtype, or a pair (a, b) standing for their channel concat (skip connection of
        the up path); tbias f32 [Cout].  Operands of the convs are compute dtype."""
        dt = self.compute_dtype
        x2 = None
        if isinstance(x, tuple):
            if dt == torch.bfloat16 and x[0].dtype == torch.float32:
                x, x2 = x       # f32 trunk halves: GroupNorm and the hi|lo split read both sources, no concat pass
            else:
                x = ops.concat_channels(*x)
        h = engine.groupnorm(x, self.groupnorm_feature, silu=True, out_dtype=dt, x2=x2)
        h = engine.conv3x3(h, self.conv_feature, bias=tbias, want_stats=True)     # feeds groupnorm_merged
        h = engine.groupnorm(h, self.groupnorm_merged, silu=True)
        skip = x if isinstance(self.residual_layer, nn.Identity) else \
            engine.conv1x1(x, self.residual_layer, out_dtype=x.dtype, compute_dtype=dt, x2=x2)
        return engine.conv3x3(h, self.conv_merged, residual=skip, want_stats=True)   # feeds the next block's GroupNorm

    def forward(self, feature, time):
        tb = self.time_bias(ops.silu(time.float().contiguous()))
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(feature, self.trunk_dtype), tb))


class UNET_AttentionBlock(engine.HipModule):
    f32_trunk = True

    def __init__(self, n_heads: int, n_embeddings: int, d_prompt=768, attention_type="xformers"):
        super().__init__()
        if attention_type not in _ATTENTION_TYPES:
            raise ValueError(f"Unknown attention type: {attention_type}. Must be 'flashattention' or 'xformers'")
        channels = n_heads * n_embeddings
        self.groupnorm = nn.GroupNorm(32, channels, eps=1e-6)
        self.conv_input = nn.Conv2d(channels, channels, kernel_size=1, padding=0)
        self.layernorm_1 = nn.LayerNorm(channels)
        self.layernorm_2 = nn.LayerNorm(channels)
        self.layernorm_3 = nn.LayerNorm(channels)
        self.attention_1 = SelfAttention(n_heads, channels, in_proj_bias=False)
        self.attention_2 = CrossAttention(n_heads, channels, d_prompt, in_proj_bias=False)
        self.linear_geglu_1 = nn.Linear(channels, 4 * channels * 2)
        self.linear_geglu_2 = nn.Linear(channels * 4, channels)
        self.conv_output = nn.Conv2d(channels, channels, kernel_size=1, padding=0)

    def prompt_kv(self, ctx):
        """K|V projection of the prompt, computed in the context's own dtype (f32 context from an
        f32 text encoder stays f32 through this once-per-prompt GEMM), stored as MFMA operands."""
        return ops.cast(engine.cross_kv(ctx, self.attention_2), self.compute_dtype)

    def run(self, x, kv):
        """x NHWC [B,h,w,C] in the trunk dtype; kv = self.prompt_kv(context) [B,77,2C].  The token
        stream ``t`` is a trunk too (three residual adds accumulate into it)."""
        B, H, W, C = x.shape
        dt, td = self.compute_dtype, x.dtype
        h = engine.groupnorm(x, self.groupnorm, silu=False, out_dtype=dt).view(B, H * W, C)
        if dt == torch.bfloat16 and td == torch.float32:
            # every producer of the token trunk also emits the LayerNorm of what it stored (one launch at C = 320)
            t, n = engine.conv1x1_ln(h, self.conv_input, self.layernorm_1)
            t, n = engine.self_attention(n, self.attention_1, residual=t, then_ln=self.layernorm_2)
            t, n = engine.cross_attention(n, self.attention_2, kv, residual=t, then_ln=self.layernorm_3)
        else:
            t = engine.conv1x1(h, self.conv_input, out_dtype=td)
            t = engine.self_attention(engine.layernorm(t, self.layernorm_1, dt), self.attention_1, residual=t)
            t = engine.cross_attention(engine.layernorm(t, self.layernorm_2, dt), self.attention_2, kv, residual=t)
            n = engine.layernorm(t, self.layernorm_3, dt)
        g = engine.linear_geglu(n, self.linear_geglu_1)
        ts = engine.linear_hilo(g, self.linear_geglu_2, t) if dt == torch.bfloat16 else None
        if ts is not None:   # t is only ever read as conv_output's hi|lo operand: the GEMM epilogue writes that form
            return engine.conv1x1_presplit(ts.view(B, H, W, 2 * C), self.conv_output, x, want_stats=True)
        t = engine.linear(g, self.linear_geglu_2, residual=t)
        return engine.conv1x1(t.view(B, H, W, C), self.conv_output, residual=x, compute_dtype=dt, want_stats=True)

    def run_cfg_shared(self, x, kv):
        """``run`` for a block whose input is the SAME for the two halves of a classifier-free-guidance batch
        (pipeline.py:147 feeds [latents; latents]; only the prompt differs): x [B,h,w,C], kv [2B,77,2C] -> [2B,h,w,C].
        Everything up to the cross attention -- GroupNorm, conv_input, LayerNorm, self attention, the cross attention's
        LayerNorm and Q projection -- depends on x alone and is computed once; the two halves split where the prompt
        enters (one cross-attention launch per half on the shared Q) and the residual adds that still read shared
        tensors run per half into the halves of the 2B result.  Same arithmetic per element as ``run`` on the
        duplicated batch."""
        B, H, W, C = x.shape
        dt, td = self.compute_dtype, x.dtype
        a2 = self.attention_2
        h = engine.groupnorm(x, self.groupnorm, silu=False, out_dtype=dt).view(B, H * W, C)
        fused_ln = dt == torch.bfloat16 and td == torch.float32 and ops.gemm_ln_fused(B * H * W, C, dt)
        if fused_ln:
            t, n = engine.conv1x1_ln(h, self.conv_input, self.layernorm_1)
            t, n = engine.self_attention(n, self.attention_1, residual=t, then_ln=self.layernorm_2)
        else:
            t = engine.conv1x1(h, self.conv_input, out_dtype=td)
            t = engine.self_attention(engine.layernorm(t, self.layernorm_1, dt), self.attention_1, residual=t)
            n = engine.layernorm(t, self.layernorm_2, dt)
        q = engine.linear(n, a2.q_proj)
        o = torch.empty((2 * B, H * W, C), dtype=q.dtype, device=q.device)
        t2 = torch.empty((2 * B, H * W, C), dtype=td, device=q.device)
        n3 = torch.empty((2 * B, H * W, C), dtype=dt, device=q.device) if fused_ln else None
        for half in range(2):
            sl = slice(half * B, (half + 1) * B)
            kvh = kv[sl]
            ops.attention(q, kvh[..., :C], kvh[..., C:], a2.n_heads, False, out=o[sl])
            if fused_ln:
                engine.linear_ln(o[sl], a2.out_proj, t, self.layernorm_3, out=t2[sl], ln_out=n3[sl])
            else:
                engine.linear(o[sl], a2.out_proj, t, out=t2[sl])
        if not fused_ln:
            n3 = engine.layernorm(t2, self.layernorm_3, dt)
        g = engine.linear_geglu(n3, self.linear_geglu_1)
        ts = engine.linear_hilo(g, self.linear_geglu_2, t2) if dt == torch.bfloat16 else None
        out = torch.empty((2 * B, H, W, C), dtype=td, device=q.device)
        if ts is None:
            t2 = engine.linear(g, self.linear_geglu_2, residual=t2)
        for half in range(2):
            sl = slice(half * B, (half + 1) * B)
            if ts is not None:
                engine.conv1x1_presplit(ts[sl].view(B, H, W, 2 * C), self.conv_output, x, out=out[sl])
            else:
                engine.conv1x1(t2[sl].view(B, H, W, C), self.conv_output, residual=x, compute_dtype=dt, out=out[sl])
        return out

    def forward(self, x, prompt):
        kv = self.prompt_kv(prompt.contiguous())
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x, self.trunk_dtype), kv))


class Upsample(engine.HipModule):
    f32_trunk = True

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, kernel_size=3, padding=1)

    def run(self, x):
        # nearest-2x folded into the conv's gather; trunk in, trunk out
        return engine.conv3x3_trunk(x, self.conv, self.compute_dtype, upsample=True)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x, self.trunk_dtype)))


class TimeEmbedding(engine.HipModule):
    def __init__(self, n_embeddings):
        super().__init__()
        self.linear_1 = nn.Linear(n_embeddings, n_embeddings * 4)
        self.linear_2 = nn.Linear(n_embeddings * 4, n_embeddings * 4)

    def forward(self, x):
        """[S,320] -> [S,1280]; always f32 (negligible cost, feeds every block's bias)."""
        x = x.float().contiguous()
        h = ops.gemm(x, engine.f32(self.linear_1.weight), engine.f32(self.linear_1.bias), act=ops.ACT_SILU)
        return ops.gemm(h, engine.f32(self.linear_2.weight), engine.f32(self.linear_2.bias))


class SwitchSequential(nn.Sequential, engine.HipModule):
    """diffusion.py:172-185: dispatches each member by type.  ``UNET.run`` drives the members itself (NHWC, hoisted
    prompt / time state); ``forward`` keeps the reference's behaviour for a caller that runs one stage directly:
    NCHW tensors in and out, every member through its own reference-signature ``forward``."""

    def forward(self, x, prompt, time):
        for layer in self:
            if isinstance(layer, UNET_AttentionBlock):
                x = layer(x, prompt)
            elif isinstance(layer, UNET_ResidualBlock):
                x = layer(x, time)
            elif isinstance(layer, Upsample):
                x = layer(x)
            else:   # bare nn.Conv2d (stem, stride-2 downsampling): parameters only, executed on the HIP kernels
                x = _run_plain_conv(layer, x, self.compute_dtype)
        return x


def _run_plain_conv(conv, x, compute_dtype):
    """A stage's bare nn.Conv2d on NCHW input, as UNET.run executes it (first conv: Cin = 4; else a trunk conv)."""
    x = x.float().contiguous()
    trunk = torch.float32     # UNet trunk tensors are f32 in both modes (f32_trunk)
    if conv.in_channels <= 4:
        return ops.nhwc_to_nchw(ops.conv3x3_cin_small(x, engine.conv3x3_cin_small_w(conv), engine.f32(conv.bias), trunk))
    return ops.nhwc_to_nchw(engine.conv3x3_trunk(ops.nchw_to_nhwc(x, trunk), conv, compute_dtype, stride=conv.stride[0]))


def _res(cin, cout):
    return [UNET_ResidualBlock(cin, cout)]


def _res_attn(cin, cout, d_head, at):
    return [UNET_ResidualBlock(cin, cout), UNET_AttentionBlock(8, d_head, attention_type=at)]


class UNET(engine.HipModule):
    f32_trunk = True

    def __init__(self, attention_type="xformers"):
        super().__init__()
        if attention_type not in _ATTENTION_TYPES:
            raise ValueError(f"Unknown attention type: {attention_type}. Must be 'flashattention' or 'xformers'")
        self.attention_type = at = attention_type
        down = lambda c: [nn.Conv2d(c, c, kernel_size=3, stride=2, padding=1)]
        enc = [[nn.Conv2d(4, 320, kernel_size=3, padding=1)],
               _res_attn(320, 320, 40, at), _res_attn(320, 320, 40, at), down(320),
               _res_attn(320, 640, 80, at), _res_attn(640, 640, 80, at), down(640),
               _res_attn(640, 1280, 160, at), _res_attn(1280, 1280, 160, at), down(1280),
               _res(1280, 1280), _res(1280, 1280)]
        self.encoders = nn.ModuleList([SwitchSequential(*s) for s in enc])
        self.bottleneck = SwitchSequential(UNET_ResidualBlock(1280, 1280), UNET_AttentionBlock(8, 160, attention_type=at),
                                           UNET_ResidualBlock(1280, 1280))
        dec = [_res(2560, 1280), _res(2560, 1280), _res(2560, 1280) + [Upsample(1280)],
               _res_attn(2560, 1280, 160, at), _res_attn(2560, 1280, 160, at),
               _res_attn(1920, 1280, 160, at) + [Upsample(1280)],
               _res_attn(1920, 640, 80, at), _res_attn(1280, 640, 80, at),
               _res_attn(960, 640, 80, at) + [Upsample(640)],
               _res_attn(960, 320, 40, at), _res_attn(640, 320, 40, at), _res_attn(640, 320, 40, at)]
        self.decoders = nn.ModuleList([SwitchSequential(*s) for s in dec])

    # -- hoistable per-prompt / per-timestep state
    def residual_blocks(self):
        return [m for m in self.modules() if isinstance(m, UNET_ResidualBlock)]

    def attention_blocks(self):
        return [m for m in self.modules() if isinstance(m, UNET_AttentionBlock)]

    def prompt_state(self, ctx):
        """K|V projections of the prompt for all 16 attention blocks (constant over the loop)."""
        ctx = ctx.contiguous()
        return {id(a): a.prompt_kv(ctx) for a in self.attention_blocks()}

    def time_bias_table(self, time):
        """time [S,1280] f32 -> f32 [S, sum(Cout)]: every block's conv bias + linear_time(silu(t)),
        one GEMM over the concatenated projections; ``time_bias_slices`` maps blocks to columns."""
        blocks = self.residual_blocks()
        w = engine.packed(tuple(b.linear_time.weight for b in blocks), "tcat", torch.float32,
                          lambda *ws: torch.cat([t.float() for t in ws], 0).contiguous())
        bias = engine.packed(tuple(p for b in blocks for p in (b.linear_time.bias, b.conv_feature.bias)), "tcatb",
                             torch.float32,
                             lambda *bs: torch.cat([(bs[2 * i] + bs[2 * i + 1]).float() for i in range(len(bs) // 2)]))
        return ops.gemm(ops.silu(time.float().contiguous()), w, bias)

    def time_bias_slices(self, row):
        """row: f32 [sum(Cout)] (one row of the table) -> {id(block): view [Cout]}."""
        out, o = {}, 0
        for b in self.residual_blocks():
            c = b.conv_feature.out_channels
            out[id(b)] = row[o:o + c]
            o += c
        return out

    def run(self, x_nchw, kv, tbias, copies=1):
        """x_nchw f32 [B,4,h,w] -> NHWC [copies*B,h,w,320] in the trunk dtype.  ``copies=2``: the UNet batch is the
        same latents twice (classifier-free guidance, pipeline.py:147); the first conv runs once and writes both."""
        dt = self.trunk_dtype
        skips = []
        x = None
        for i, stage in enumerate(self.encoders):
            if i == 0:
                c = stage[0]
                x = ops.conv3x3_cin_small(x_nchw, engine.conv3x3_cin_small_w(c), engine.f32(c.bias), dt, copies)
            elif i == 1 and copies == 2 and SHARE_CFG_PREFIX:
                # both halves of the batch are still the same tensor here: the first residual block and the first
                # attention block up to its cross attention run once, on one half (UNET_AttentionBlock.run_cfg_shared)
                res, att = stage[0], stage[1]
                x = att.run_cfg_shared(res.run(x[:x.shape[0] // 2], tbias[id(res)]), kv[id(att)])
            else:
                x = self._stage(stage, x, kv, tbias)
            skips.append(x)
        x = self._stage(self.bottleneck, x, kv, tbias)
        for stage in self.decoders:
            x = self._stage(stage, (x, skips.pop()), kv, tbias)   # torch.cat((x, skip), dim=1), diffusion.py:279: never materialised
        return x

    def _stage(self, stage, x, kv, tbias):
        for m in stage:
            if isinstance(m, UNET_ResidualBlock):
                x = m.run(x, tbias[id(m)])
            elif isinstance(m, UNET_AttentionBlock):
                x = m.run(x, kv[id(m)])
            elif isinstance(m, Upsample):
                x = m.run(x)
            else:  # stride-2 downsampling conv: trunk in, trunk out
                x = engine.conv3x3_trunk(x, m, self.compute_dtype, stride=2)
        return x

    def forward(self, x, prompt, time):
        """x [B,4,h,w], prompt [B,77,768], time [1,1280] -> [B,320,h,w] (diffusion.py:264-281)."""
        dt = self.compute_dtype
        kv = self.prompt_state(prompt)
        tb = self.time_bias_slices(self.time_bias_table(time).reshape(-1))
        return ops.nhwc_to_nchw(self.run(x.float().contiguous(), kv, tb))


class UNET_Output_Layer(engine.HipModule):
    f32_trunk = True

    def __init__(self, input_channel, output_channel):
        super().__init__()
        self.groupnorm = nn.GroupNorm(32, input_channel)
        self.conv = nn.Conv2d(input_channel, output_channel, kernel_size=3, padding=1)

    def run(self, x):
        """NHWC [B,h,w,320] (trunk dtype) -> NCHW f32 [B,4,h,w].  Stays in the trunk dtype: nothing
        dilutes this layer's rounding (its output IS eps, which CFG then amplifies), and it is
        0.006 % of the UNet's FLOPs."""
        h = engine.groupnorm(x, self.groupnorm, silu=True)
        return ops.conv3x3_cout_small(h, engine.conv3x3_cout_small_w(self.conv, h.dtype), engine.f32(self.conv.bias),
                                      self.conv.out_channels)

    def forward(self, x):
        return self.run(ops.nchw_to_nhwc(x, self.trunk_dtype))


class Diffusion(engine.HipModule):
    f32_trunk = True

    def __init__(self, attention_type="xformers"):
        super().__init__()
        self.time_embedding = TimeEmbedding(320)
        self.unet = UNET(attention_type=attention_type)
        self.final = UNET_Output_Layer(320, 4)

    def run(self, latent, kv, tbias, copies=1):
        """latent NCHW f32 [B,4,h,w] -> eps NCHW f32 [copies*B,4,h,w] with hoisted prompt/time state."""
        return self.final.run(self.unet.run(latent, kv, tbias, copies))

    def forward(self, latent, prompt, time):
        """latent [B,4,h,w], prompt [B,77,768], time [1,320] -> [B,4,h,w] f32 (diffusion.py:309-324)."""
        dt = self.compute_dtype
        kv = self.unet.prompt_state(prompt)
        table = self.unet.time_bias_table(self.time_embedding(time))
        return self.run(latent.float().contiguous(), kv, self.unet.time_bias_slices(table.reshape(-1)))
