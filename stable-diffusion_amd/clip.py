"""CLIP ViT-L/14 text encoder behind the reference's ``CLIP`` interface (clip.py:7-97):
token+position embedding gather, 12 pre-LN layers (causal flash attention over 77 tokens,
quick-GELU MLP fused into the first MLP GEMM's epilogue), final LayerNorm."""
import torch
import torch.nn as nn

from . import engine, ops
from .attention import SelfAttention

_ATTENTION_TYPES = ("xformers", "flashattention", "default")


class CLIPEmbedding(engine.HipModule):
    f32_trunk = True

    def __init__(self, vocab_size, n_embeddings, n_tokens: int):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab_size, n_embeddings)
        self.position_embedding = nn.Parameter(torch.zeros(n_tokens, n_embeddings))

    def forward(self, tokens):
        dt = self.trunk_dtype
        table = engine.packed(self.token_embedding.weight, "emb", dt, lambda w: w.to(dt).contiguous())
        pos = engine.packed(self.position_embedding, "pos", dt, lambda w: w.to(dt).contiguous())
        return ops.clip_embed(tokens.to(torch.long).contiguous(), table, pos)


class CLIPLayer(engine.HipModule):
    f32_trunk = True

    def __init__(self, n_heads: int, n_embeddings: int, attention_type="xformers"):
        super().__init__()
        if attention_type not in _ATTENTION_TYPES:
            raise ValueError(f"Unknown attention type: {attention_type}")
        self.layernorm_1 = nn.LayerNorm(n_embeddings)
        self.attention = SelfAttention(n_heads, n_embeddings)
        self.layernorm_2 = nn.LayerNorm(n_embeddings)
        self.linear_1 = nn.Linear(n_embeddings, n_embeddings * 4)
        self.linear_2 = nn.Linear(n_embeddings * 4, n_embeddings)

    def run(self, x):
        """x [B,77,768] in the trunk dtype."""
        dt = self.compute_dtype
        x = engine.self_attention(engine.layernorm(x, self.layernorm_1, dt), self.attention, causal=True, residual=x)
        h = engine.linear(engine.layernorm(x, self.layernorm_2, dt), self.linear_1, act=ops.ACT_QUICK_GELU)
        return engine.linear(h, self.linear_2, residual=x)

    def forward(self, x):
        return self.run(ops.cast(x.contiguous(), self.trunk_dtype))


class CLIP(engine.HipModule):
    f32_trunk = True

    def __init__(self, attention_type="xformers"):
        super().__init__()
        self.embedding = CLIPEmbedding(49408, 768, 77)
        self.layers = nn.ModuleList([CLIPLayer(12, 768, attention_type=attention_type) for _ in range(12)])
        self.layernorm = nn.LayerNorm(768)

    def forward(self, tokens):
        """tokens int64 [B,77] -> [B,77,768] in compute dtype (clip.py:86-97)."""
        state = self.embedding(tokens)
        for layer in self.layers:
            state = layer.run(state)
        return engine.layernorm(state, self.layernorm, self.compute_dtype)
