"""Where the halo-staged conv's time goes: full / no DMA (debug 1) / no MFMA (debug 2), hot and HBM-cold operands,
against the gathered form.  Timing only (the ablations compute garbage)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()
OFF = 8 << 20
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1)
def t(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(n): fn(i)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / n)
    return sorted(ts)[2]
for B, H, W, Cin, Cout in [(16, 64, 64, 320, 320), (16, 64, 64, 960, 320), (16, 32, 32, 1920, 640), (16, 16, 16, 2560, 1280)]:
    for NR in (1, 8):
        xs = [rnd(B, H, W, Cin).bfloat16() for _ in range(NR)]
        w, b = (rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5)).bfloat16(), torch.rand(Cout, device=dev, generator=g)
        row = []
        HBW = 18 << 8; HALO = 2 << 20
        for nm, impl in (("gather", OFF), ("halo", HALO), ("hbw", 0)):
            L.sdamd_set_gemm_impl(impl)
            ms = t(lambda i: ops.conv3x3(xs[i % NR], w, b), 8)
            row.append(f"{nm} {ms*1e3:6.1f}us {2.0*B*H*W*Cout*9*Cin/ms/1e9:5.0f}TF")
        print(f"conv B{B} {H}x{W} {Cin}->{Cout} {'hot ' if NR == 1 else 'cold'}: " + " | ".join(row), flush=True)
L.sdamd_set_gemm_impl(0)
