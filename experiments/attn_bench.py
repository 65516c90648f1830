"""Attention kernel timing on the UNet's shapes (batch 16), random data (experiment)."""
import sys; sys.path.insert(0, '.')
import torch
import sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
import sys as _s
from sd_amd import _lib
_lib.lib().sdamd_set_gemm_impl((int(_s.argv[1]) if len(_s.argv) > 1 else 0) << 8)
shapes = [(16, 8, 40, 4096, 4096), (16, 8, 80, 1024, 1024), (16, 8, 160, 256, 256), (16, 8, 40, 4096, 77), (16, 8, 80, 1024, 77), (16, 8, 160, 256, 77), (16, 8, 160, 64, 64)]
for B, H, d, Sq, Skv in shapes:
    C = H * d
    qkv = rnd(B, Sq, 3 * C) if Sq == Skv else None
    q = qkv[..., :C] if qkv is not None else rnd(B, Sq, C)
    k = qkv[..., C:2 * C] if qkv is not None else rnd(B, Skv, 2 * C)[..., :C]
    v = qkv[..., 2 * C:] if qkv is not None else rnd(B, Skv, 2 * C)[..., C:]
    for _ in range(3): ops.attention(q, k, v, H)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4): ops.attention(q, k, v, H)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 4)
    t = sorted(ts)[2]
    print(f"attn B{B} H{H} d{d} Sq{Sq} Skv{Skv}: {t*1e3:8.1f} us  {4.0*B*H*Sq*Skv*d/t/1e9:7.1f} TF", flush=True)
