"""Phase-form upsample conv on the big-wave halo kernel (gemm5.hip) against the gathered kernel: time, rotated operands."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()
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
for B, HW, Cin, Cout in [(16, 32, 1280, 640), (16, 16, 2560, 1280), (2, 32, 1280, 640)]:
    NR = 4
    xs = [rnd(B, HW, HW, Cin).bfloat16() for _ in range(NR)]
    w, b = (rnd(4, Cout, 4 * Cin) * ((4 * Cin) ** -0.5)).bfloat16(), torch.rand(Cout, device=dev, generator=g)
    row, outs = [], {}
    for nm, impl in (("gather", 8 << 20), ("hbw", 0)):
        L.sdamd_set_gemm_impl(impl)
        outs[nm] = ops.conv3x3(xs[0], w, b, None, 1, 0, 2, out_f32=True)
        ms = t(lambda i: ops.conv3x3(xs[i % NR], w, b, None, 1, 0, 2, out_f32=True), 2 * NR)
        row.append(f"{nm} {ms*1e3:6.1f}us {2.0*4*B*HW*HW*Cout*4*Cin/ms/1e9:5.0f}TF executed")
    d = (outs["hbw"] - outs["gather"]).abs().max().item()
    print(f"up conv B{B} {HW}x{HW} {Cin}->{Cout}: " + " | ".join(row) + f" | max diff {d:.2e}", flush=True)
L.sdamd_set_gemm_impl(0)
