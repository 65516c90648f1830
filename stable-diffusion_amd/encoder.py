"""VAE encoder behind the reference's ``VAE_Encoder`` interface (encoder.py:6-86), img2img only.
Stride-2 convs use the bottom/right-only padding of encoder.py:65-66 inside the conv kernel; the
final 1x1 conv, clamp/exp/sqrt, reparameterisation with the caller's noise and the 0.18215 scale
are one kernel."""
import torch.nn as nn

from . import engine, ops
from .decoder import VAE_AttentionBlock, VAE_ResidualBlock, run_vae_stack


class VAE_Encoder(nn.Sequential, engine.HipModule):
    f32_trunk = True

    def __init__(self):
        r = VAE_ResidualBlock
        super().__init__(
            nn.Conv2d(3, 128, kernel_size=3, padding=1),
            r(128, 128), r(128, 128),
            nn.Conv2d(128, 128, kernel_size=3, stride=2, padding=0),
            r(128, 256), r(256, 256),
            nn.Conv2d(256, 256, kernel_size=3, stride=2, padding=0),
            r(256, 512), r(512, 512),
            nn.Conv2d(512, 512, kernel_size=3, stride=2, padding=0),
            r(512, 512), r(512, 512), r(512, 512),
            VAE_AttentionBlock(512),
            r(512, 512),
            nn.GroupNorm(32, 512), nn.SiLU(),
            nn.Conv2d(512, 8, kernel_size=3, padding=1),
            nn.Conv2d(8, 8, kernel_size=1, padding=0),
        )

    def forward(self, x, noise):
        """x [B,3,H,W] f32 in (-1,1), noise [B,4,H/8,W/8] -> latents [B,4,H/8,W/8] f32."""
        dt = self.compute_dtype
        mods = list(self)
        h = ops.conv3x3_cin_small(x.float().contiguous(), engine.conv3x3_cin_small_w(mods[0]), engine.f32(mods[0].bias),
                                  self.trunk_dtype)
        h = run_vae_stack(mods[1:15], h, dt)
        h = engine.groupnorm(h, mods[15], silu=True, out_dtype=dt)
        moments = ops.conv3x3_cout_small(h, engine.conv3x3_cout_small_w(mods[17], dt), engine.f32(mods[17].bias), 8)
        return ops.vae_sample(moments, noise.float().contiguous(), engine.f32(mods[18].weight), engine.f32(mods[18].bias))
