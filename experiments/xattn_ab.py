"""Cross attention (77 keys) per launch: the K/V-stationary kernel (attention3.hip) against the generic flash kernel
(debug word 26), at the UNet's shapes at batch 16.  Usage: python experiments/xattn_ab.py"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("stable-diffusion_amd")
from sd_amd import ops, _lib
L = _lib.lib()
DEV = "cuda"

def timeit(fn, n=40):
    """n launches captured in one graph (the host side of an eager call is ~20 us, more than the small launches)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3

for (B, H, d, Sq) in ((16, 8, 40, 4096), (8, 8, 40, 4096), (16, 8, 80, 1024), (16, 8, 160, 256), (16, 8, 160, 64), (2, 8, 40, 4096)):
    C = H * d
    g = torch.Generator().manual_seed(0)
    qq = torch.randn(B, Sq, C, generator=g).bfloat16().to(DEV)
    kv = torch.randn(B, 77, 2 * C, generator=g).bfloat16().to(DEV)
    f = lambda: ops.attention(qq, kv[..., :C], kv[..., C:], H)
    t_new = timeit(f)
    a = f()
    L.sdamd_set_gemm_impl(26 << 8)
    t_old = timeit(f)
    b = f()
    L.sdamd_set_gemm_impl(0)
    err = ((a.float() - b.float()).norm() / b.float().norm()).item()
    byts = 2 * B * Sq * C * 2
    print(f"B{B} H{H} d{d} Sq{Sq} Skv77: stationary {t_new:7.1f} us ({byts / t_new / 1e6:5.2f} TB/s)   generic {t_old:7.1f} us   rel-L2 {err:.1e}", flush=True)
