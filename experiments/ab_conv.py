"""Two conv shapes + two GEMM shapes, for A/B between library builds (SDAMD_LIB)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 4)
    return sorted(ts)[2]
out = []
for B, H, W, Cin, Cout in [(16, 64, 64, 640, 320), (16, 32, 32, 640, 640), (16, 16, 16, 2560, 1280)]:
    x, w, b = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), torch.rand(Cout, device=dev, generator=g)
    ms = t(lambda: ops.conv3x3(x, w, b, None, 1, 0, False))
    out.append(f"conv{H}x{Cin}->{Cout} {2.0*B*H*W*Cout*9*Cin/ms/1e9:6.0f}TF")
for M, N, K in [(16384, 640, 2560), (4096, 3840, 1280), (16384, 5120, 640), (65536, 960, 320)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    ms = t(lambda: ops.gemm(a, w, None, None))
    out.append(f"gemm{M}x{N}x{K} {2.0*M*N*K/ms/1e9:6.0f}TF")
print("  ".join(out))
