import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"; L = _lib.lib()
def t(call, iters=10):
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K) in ((65536, 2560, 64), (65536, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
    call = lambda: ops.gemm(a, w, b, None, ops.ACT_GEGLU)
    L.sdamd_set_gemm_impl(0); full = t(call)
    L.sdamd_set_gemm_impl(7 << 8); nost = t(call)
    L.sdamd_set_gemm_impl(0)
    print(f"M{M} N{N} K{K}: full {full:.1f} us   without the global stores {nost:.1f} us", flush=True)
