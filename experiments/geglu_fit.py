"""Fixed (K-independent) cost of the fused-GeGLU and plain 256x256 / 128x320 launches: time vs K at fixed M, N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
def t(call, iters=10):
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, geglu, res) in ((65536, 2560, True, False), (65536, 960, False, False), (65536, 320, False, True)):
    row = []
    for K in (64, 128, 320, 640, 1280, 2560):
        a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if res else None
        us = t(lambda: ops.gemm(a, w, b, r, ops.ACT_GEGLU if geglu else ops.ACT_NONE, out_f32=res))
        row.append(f"K{K}: {us:.1f}")
    print(f"M{M} N{N} geglu={geglu} res={res}: " + "  ".join(row), flush=True)
