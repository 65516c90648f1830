"""Fused-GeGLU launches: 256x256 ping-pong tile (1 workgroup per CU) vs the 128x128 4-wave tile (2 per CU), per UNet level,
operands rotated over buffer sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
L = _lib.lib()
def t(call, sets, iters=9, rounds=3):
    ts = []
    for _ in range(rounds):
        call(*sets[0]); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters): call(*sets[i % len(sets)])
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(ts)[1]
for (M, N, K) in ((65536, 2560, 320), (32768, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280), (8192, 2560, 320)):
    sets = []
    for _ in range(3):
        a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
        sets.append((a, w, b))
    call = lambda a, w, b: ops.gemm(a, w, b, None, ops.ACT_GEGLU)
    L.sdamd_set_gemm_impl(0); t256 = t(call, sets)
    L.sdamd_set_gemm_impl(6 << 8); t128 = t(call, sets)
    L.sdamd_set_gemm_impl(0)
    print(f"M{M} N{N} K{K}: 256x256 pp {t256:.1f} us   128x128 {t128:.1f} us   ({100 * (t128 / t256 - 1):+.1f} %)", flush=True)
