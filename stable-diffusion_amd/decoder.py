"""VAE decoder behind the reference's ``VAE_Decoder`` interface (decoder.py:6-134).

nn.Sequential child indices ("0." ... "25.") are the weights ABI.  Execution is NHWC: the three
nn.Upsample stages are folded into the following conv's gather, GN+SiLU is one fused pass, the
512-wide single-head attention is a flash kernel of its own (attn512_kernel: no [S, S] tensor), and the last conv writes
the caller's NCHW f32 image directly.  Unlike the reference (decoder.py:128) the caller's latent
tensor is NOT modified in place.
"""
import torch
import torch.nn as nn

from . import engine, ops
from .attention import SelfAttention


class VAE_ResidualBlock(engine.HipModule):
    f32_trunk = True

    def __init__(self, input_channel, output_channel):
        super().__init__()
        self.groupnorm_1 = nn.GroupNorm(32, input_channel)
        self.conv_1 = nn.Conv2d(input_channel, output_channel, kernel_size=3, padding=1)
        self.groupnorm_2 = nn.GroupNorm(32, output_channel)
        self.conv_2 = nn.Conv2d(output_channel, output_channel, kernel_size=3, padding=1)
        self.residual_layer = (nn.Identity() if input_channel == output_channel
                               else nn.Conv2d(input_channel, output_channel, kernel_size=1, padding=0))

    def run(self, x):
        """x NHWC in the trunk dtype (f32 trunk / compute-dtype operands, as in the UNet blocks)."""
        dt = self.compute_dtype
        h = engine.conv3x3(engine.groupnorm(x, self.groupnorm_1, silu=True, out_dtype=dt), self.conv_1)
        h = engine.groupnorm(h, self.groupnorm_2, silu=True)
        skip = x if isinstance(self.residual_layer, nn.Identity) else \
            engine.conv1x1(ops.cast(x, dt), self.residual_layer, out_dtype=x.dtype)
        return engine.conv3x3(h, self.conv_2, residual=skip)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x, self.trunk_dtype)))


class VAE_AttentionBlock(engine.HipModule):
    f32_trunk = True

    def __init__(self, channels):
        super().__init__()
        self.groupnorm = nn.GroupNorm(32, channels)
        self.attention = SelfAttention(1, channels)

    def run(self, x):
        B, H, W, C = x.shape
        t = engine.groupnorm(x, self.groupnorm, silu=False, out_dtype=self.compute_dtype).view(B, H * W, C)
        return engine.self_attention(t, self.attention, residual=x.view(B, H * W, C)).view(B, H, W, C)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x, self.trunk_dtype)))


def run_vae_stack(modules, x, compute_dtype):
    """Drive decoder/encoder members on an NHWC trunk tensor.  The plain convs between blocks take
    the trunk as operand: it is rounded to the compute dtype for them (no CFG after the VAE, so the
    two-term split the UNet uses there is not worth its FLOPs)."""
    pending_up = False
    for m in modules:
        if isinstance(m, (VAE_ResidualBlock, VAE_AttentionBlock)):
            x = m.run(x)
        elif isinstance(m, nn.Upsample):
            pending_up = True
        elif isinstance(m, nn.Conv2d):
            if m.stride == (2, 2):   # encoder: F.pad(0,1,0,1) + stride-2 conv (encoder.py:65-66)
                x = engine.conv3x3(ops.cast(x, compute_dtype), m, stride=2, pad_mode=1, out_dtype=x.dtype)
            else:
                x = engine.conv3x3(ops.cast(x, compute_dtype), m, upsample=pending_up, out_dtype=x.dtype)
            pending_up = False
        else:
            raise TypeError(type(m))
    return x


class VAE_Decoder(nn.Sequential, engine.HipModule):
    f32_trunk = True

    def __init__(self):
        r = VAE_ResidualBlock
        super().__init__(
            nn.Conv2d(4, 4, kernel_size=1, padding=0),
            nn.Conv2d(4, 512, kernel_size=3, padding=1),
            r(512, 512), VAE_AttentionBlock(512), r(512, 512), r(512, 512), r(512, 512), r(512, 512),
            nn.Upsample(scale_factor=2), nn.Conv2d(512, 512, kernel_size=3, padding=1),
            r(512, 512), r(512, 512), r(512, 512),
            nn.Upsample(scale_factor=2), nn.Conv2d(512, 512, kernel_size=3, padding=1),
            r(512, 256), r(256, 256), r(256, 256),
            nn.Upsample(scale_factor=2), nn.Conv2d(256, 256, kernel_size=3, padding=1),
            r(256, 128), r(128, 128), r(128, 128),
            nn.GroupNorm(32, 128), nn.SiLU(), nn.Conv2d(128, 3, kernel_size=3, padding=1),
        )

    def forward(self, x):
        """latents [B,4,h,w] f32 -> image [B,3,8h,8w] f32 (decoder.py:126-134)."""
        dt = self.compute_dtype
        mods = list(self)
        pq = ops.vae_post_quant(x.float().contiguous(), engine.f32(mods[0].weight), engine.f32(mods[0].bias))
        h = ops.conv3x3_cin_small(pq, engine.conv3x3_cin_small_w(mods[1]), engine.f32(mods[1].bias), self.trunk_dtype)
        h = run_vae_stack(mods[2:23], h, dt)
        h = engine.groupnorm(h, mods[23], silu=True, out_dtype=dt)
        last = mods[25]
        return ops.conv3x3_cout_small(h, engine.conv3x3_cout_small_w(last, dt), engine.f32(last.bias), 3)
