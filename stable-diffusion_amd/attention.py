"""SelfAttention / CrossAttention with the reference's constructor and forward signatures
(attention.py:16-133 of the reference) executing on the MFMA flash-attention kernel.

Inputs are token-major [B, S, D] HIP tensors (float32 or bfloat16); the result has the input's
dtype.  ``FlashSelfAttention`` / ``FlashCrossAttention`` (flash_attention.py:25-132) are aliases:
the reference's Triton path is replaced outright, there is only this one implementation.
"""
import torch.nn as nn

from . import engine, ops


class SelfAttention(engine.HipModule):
    def __init__(self, n_heads: int, d_embeddings: int, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.in_proj = nn.Linear(d_embeddings, 3 * d_embeddings, bias=in_proj_bias)
        self.out_proj = nn.Linear(d_embeddings, d_embeddings, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embeddings // n_heads

    def forward(self, x, causal_mask=False):
        return engine.self_attention(_tokens(x), self, causal_mask)


class CrossAttention(engine.HipModule):
    def __init__(self, n_heads, d_embeddings: int, d_cross: int, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.q_proj = nn.Linear(d_embeddings, d_embeddings, bias=in_proj_bias)
        self.k_proj = nn.Linear(d_cross, d_embeddings, bias=in_proj_bias)
        self.v_proj = nn.Linear(d_cross, d_embeddings, bias=in_proj_bias)
        self.out_proj = nn.Linear(d_embeddings, d_embeddings, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embeddings // n_heads

    def forward(self, x, y):
        x = _tokens(x)
        kv = engine.cross_kv(ops.cast(_tokens(y), x.dtype), self)
        return engine.cross_attention(x, self, kv)


def _tokens(x):
    if x.dim() != 3:
        raise ValueError(f"expected [batch, seq, dim], got {tuple(x.shape)}")
    return x.contiguous()
