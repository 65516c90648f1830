"""128x160 tile: 2-stage vs 3-stage LDS ring on the small-grid shapes (16x16 / 8x8 levels at UNet batch 16, everything at batch 2)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
L = _lib.lib()
def force(idx, splits):
    L.sdamd_set_gemm_impl(0 if idx is None else (32 + 8 * idx + splits) << 8)
def t(fn, rounds=7, inner=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / inner)
    return sorted(ts)[len(ts) // 2]
print("gemm")
for M, N, K, res in [(4096, 1280, 1280, True), (4096, 1280, 1280, False), (4096, 3840, 1280, False), (4096, 1280, 5120, True), (4096, 1280, 2560, True),
                     (1024, 1280, 1280, True), (1024, 1280, 5120, True), (16384, 640, 640, True), (16384, 640, 2560, True), (65536, 320, 320, True)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = torch.rand(M, N, device=dev, generator=g) if res else None
    row = []
    for splits in (1, 2, 4):
        for idx, nm in ((0, "s2"), (5, "s3")):
            if splits > 1 and K // 64 // splits < 4: continue
            force(idx, splits)
            ms = t(lambda: ops.gemm(a, w, None, r, out_f32=res))
            row.append(f"k{splits}{nm} {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:4.0f}TF")
    force(None, 0)
    ms = t(lambda: ops.gemm(a, w, None, r, out_f32=res))
    print(f"  M{M} N{N} K{K} res={int(res)}  auto {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:4.0f}TF | " + "  ".join(row))
print("conv")
for B, H, Cin, Cout, res in [(16, 16, 1280, 1280, True), (16, 16, 2560, 1280, False), (16, 8, 1280, 1280, True), (16, 8, 2560, 1280, False), (16, 32, 640, 640, True), (2, 64, 320, 320, True), (2, 32, 640, 640, True)]:
    x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), torch.rand(Cout, device=dev, generator=g)
    r = torch.rand(B, H, H, Cout, device=dev, generator=g) if res else None
    row = []
    for splits in (1, 2, 4, 7):
        for idx, nm in ((0, "s2"), (5, "s3")):
            force(idx, splits)
            ms = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False, out_f32=res))
            row.append(f"k{splits}{nm} {ms*1e3:6.1f}us {2.0*B*H*H*Cout*9*Cin/ms/1e9:4.0f}TF")
    force(None, 0)
    ms = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False, out_f32=res))
    print(f"  B{B} H{H} {Cin}->{Cout} res={int(res)}  auto {ms*1e3:6.1f}us {2.0*B*H*H*Cout*9*Cin/ms/1e9:4.0f}TF | " + "  ".join(row))
