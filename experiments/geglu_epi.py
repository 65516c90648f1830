"""GeGLU epilogue cost: fused GeGLU launch vs the same GEMM with the plain bf16 epilogue (twice the output bytes)."""
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
M, N = 65536, 2560
for K in (64, 320, 1280):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
    g = t(lambda: ops.gemm(a, w, b, None, ops.ACT_GEGLU))
    p = t(lambda: ops.gemm(a, w, b, None, ops.ACT_NONE))
    print(f"K{K}: geglu {g:.1f} us   plain bf16 out (2x bytes) {p:.1f} us", flush=True)
