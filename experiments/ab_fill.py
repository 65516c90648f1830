"""Is the bf16 conv kernel bound by the L2->LDS fill of the activation operand?  Timing ablation (library built with
-DSDAMD_FILL_ABLATE, wrong results): debug 3 fetches the activation tile for tap (0,0) only, i.e. A fill / 9, same
MFMA and ds_read work.  run: SDAMD_LIB=experiments/ab/libsdamd_fill.so python experiments/ab_fill.py"""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 4)
    return sorted(ts)[3]
for B, H, W, Cin, Cout in [(16, 64, 64, 320, 320), (16, 64, 64, 640, 320), (16, 32, 32, 640, 640), (16, 32, 32, 1280, 640), (16, 16, 16, 1280, 1280), (16, 16, 16, 2560, 1280)]:
    x, w, b = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), torch.rand(Cout, device=dev, generator=g)
    row = []
    for dbg, nm in ((0, "full"), (3, "A fill/9"), (1, "no DMA"), (2, "no MFMA")):
        L.sdamd_set_gemm_impl(dbg << 8)
        ms = t(lambda: ops.conv3x3(x, w, b, None, 1, 0, False))
        row.append(f"{nm} {ms*1e3:6.1f}us {2.0*B*H*W*Cout*9*Cin/ms/1e9:5.0f}TF")
    print(f"conv {H}x{W} {Cin}->{Cout}: " + " | ".join(row), flush=True)
L.sdamd_set_gemm_impl(0)
