"""PP conv kernel: full vs no-DMA (debug 1) vs no-MFMA (debug 2) timing (wrong results in the ablations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
L = _lib.lib()
def t(call, iters=10):
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, H, Cin, Cout) in ((8, 64, 512, 512), (8, 128, 256, 256)):
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).bfloat16()
    b = torch.randn(Cout, device="cuda")
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    row = []
    for pp in (0, 1):
        for dbg in ((0, 1, 2) if pp == 0 else (0, 1, 2, 3, 5)):
            L.sdamd_set_gemm_impl((pp << 20) | (dbg << 8))
            us = t(lambda: ops.conv3x3(x, w, b))
            row.append(f"pp{pp} dbg{dbg}: {us:7.1f} us {fl / us / 1e6:6.0f} TF")
    L.sdamd_set_gemm_impl(0)
    print(f"conv B{B} {H}^2 {Cin}->{Cout}: " + " | ".join(row), flush=True)
