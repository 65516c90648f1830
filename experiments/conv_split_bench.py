"""Forced tile / split-K sweep for the low-resolution conv and GEMM shapes (experiment)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"; L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
TILES = ["128x160", "128x320", "256x256", "256x160"]
cfgs = [(0, 1), (0, 2), (0, 4), (1, 1), (1, 2), (1, 4), (1, 8)]
def run(fn, fl, label):
    out = []
    for tile, s in cfgs:
        L.sdamd_set_gemm_impl((32 + 8 * tile + s) << 8)
        for _ in range(2): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); [fn() for _ in range(4)]; b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 4)
        t = sorted(ts)[2]
        out.append(f"{TILES[tile]}/S{s} {t*1e3:6.1f}us {fl/t/1e9:6.0f}TF")
    L.sdamd_set_gemm_impl(0)
    print(label, " | ".join(out), flush=True)
for B, H, W, Cin, Cout in [(16, 16, 16, 1280, 1280), (16, 16, 16, 2560, 1280), (16, 8, 8, 1280, 1280), (16, 8, 8, 2560, 1280), (16, 32, 32, 640, 640)]:
    x, w = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5)
    r = torch.rand(B, H, W, Cout, device=dev, generator=g); bias = torch.rand(Cout, device=dev, generator=g)
    run(lambda: ops.conv3x3(x, w, bias, r, out_f32=True), 2.0 * B * H * W * Cout * 9 * Cin, f"conv {H}x{W} {Cin}->{Cout}:")
for M, N, K in [(4096, 1280, 5120), (4096, 1280, 1280), (4096, 3840, 1280), (1024, 1280, 5120), (16384, 640, 2560), (16384, 640, 640)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5); r = torch.rand(M, N, device=dev, generator=g)
    run(lambda: ops.gemm(a, w, None, r, out_f32=True), 2.0 * M * N * K, f"gemm M{M} N{N} K{K}:")
