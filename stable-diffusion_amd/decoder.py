"""VAE decoder behind the reference's ``VAE_Decoder`` interface (decoder.py:6-134).

nn.Sequential child indices ("0." ... "25.") are the weights ABI.  Execution is NHWC: the three
nn.Upsample stages are folded into the following conv's gather, GN+SiLU is one fused pass, the
512-wide single-head attention runs as MFMA GEMMs around a row softmax, and the last conv writes
the caller's NCHW f32 image directly.  Unlike the reference (decoder.py:128) the caller's latent
tensor is NOT modified in place.
"""
import torch
import torch.nn as nn

from . import engine, ops
from .attention import SelfAttention


class VAE_ResidualBlock(engine.HipModule):
    def __init__(self, input_channel, output_channel):
        super().__init__()
        self.groupnorm_1 = nn.GroupNorm(32, input_channel)
        self.conv_1 = nn.Conv2d(input_channel, output_channel, kernel_size=3, padding=1)
        self.groupnorm_2 = nn.GroupNorm(32, output_channel)
        self.conv_2 = nn.Conv2d(output_channel, output_channel, kernel_size=3, padding=1)
        self.residual_layer = (nn.Identity() if input_channel == output_channel
                               else nn.Conv2d(input_channel, output_channel, kernel_size=1, padding=0))

    def run(self, x):
        h = engine.conv3x3(engine.groupnorm(x, self.groupnorm_1, silu=True), self.conv_1)
        h = engine.groupnorm(h, self.groupnorm_2, silu=True)
        skip = x if isinstance(self.residual_layer, nn.Identity) else engine.conv1x1(x, self.residual_layer)
        return engine.conv3x3(h, self.conv_2, residual=skip)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x, self.compute_dtype)))


class VAE_AttentionBlock(engine.HipModule):
    def __init__(self, channels):
        super().__init__()
        self.groupnorm = nn.GroupNorm(32, channels)
        self.attention = SelfAttention(1, channels)

    def run(self, x):
        B, H, W, C = x.shape
        t = engine.groupnorm(x, self.groupnorm, silu=False).view(B, H * W, C)
        return engine.self_attention(t, self.attention, residual=x.view(B, H * W, C)).view(B, H, W, C)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x, self.compute_dtype)))


def run_vae_stack(modules, x):
    """Drive decoder/encoder members on an NHWC tensor; returns (x, index of the first member
    not consumed) so the callers can finish with their own epilogue."""
    pending_up = False
    for m in modules:
        if isinstance(m, (VAE_ResidualBlock, VAE_AttentionBlock)):
            x = m.run(x)
        elif isinstance(m, nn.Upsample):
            pending_up = True
        elif isinstance(m, nn.Conv2d):
            if m.stride == (2, 2):   # encoder: F.pad(0,1,0,1) + stride-2 conv (encoder.py:65-66)
                x = engine.conv3x3(x, m, stride=2, pad_mode=1)
            else:
                x = engine.conv3x3(x, m, upsample=pending_up)
            pending_up = False
        else:
            raise TypeError(type(m))
    return x


class VAE_Decoder(nn.Sequential, engine.HipModule):
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
        h = ops.conv3x3_cin_small(pq, engine.conv3x3_cin_small_w(mods[1]), engine.f32(mods[1].bias), dt)
        h = run_vae_stack(mods[2:23], h)
        h = engine.groupnorm(h, mods[23], silu=True)
        last = mods[25]
        return ops.conv3x3_cout_small(h, engine.conv3x3_cout_small_w(last, dt), engine.f32(last.bias), 3)
