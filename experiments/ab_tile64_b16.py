"""UNet batch 16, the 8x8 and 16x16 levels: auto plan (128x160 + split-K) vs the 64x160 tile with fewer splits."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
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
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
for B, H, Cin, Cout, res in [(16, 8, 1280, 1280, True), (16, 8, 2560, 1280, False), (16, 16, 1280, 1280, True), (16, 16, 2560, 1280, False), (16, 16, 640, 1280, False)]:
    x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), torch.rand(Cout, device=dev, generator=g)
    r = torch.rand(B, H, H, Cout, device=dev, generator=g) if res else None
    force(None, 0)
    ms = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False, out_f32=res))
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    row = [f"auto {ms*1e3:6.1f}us {fl/ms/1e9:4.0f}TF |"]
    for idx, nm in ((0, "128"), (6, "64")):
        for splits in (1, 2, 3, 4, 6, 7):
            force(idx, splits)
            ms = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False, out_f32=res))
            row.append(f"{nm}k{splits} {ms*1e3:5.1f}")
    print(f"  B{B} H{H} {Cin}->{Cout} res={int(res)}  " + " ".join(row))
for M, N, K, res in [(1024, 1280, 1280, True), (1024, 1280, 5120, True), (1024, 10240, 1280, False), (4096, 1280, 1280, True), (4096, 1280, 5120, True), (4096, 1280, 2560, True), (4096, 3840, 1280, False)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = torch.rand(M, N, device=dev, generator=g) if res else None
    force(None, 0)
    ms = t(lambda: ops.gemm(a, w, None, r, out_f32=res))
    row = [f"auto {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:4.0f}TF |"]
    for idx, nm in ((0, "128"), (6, "64")):
        for splits in (1, 2, 4):
            if splits > 1 and K // 64 // splits < 4: continue
            force(idx, splits)
            ms = t(lambda: ops.gemm(a, w, None, r, out_f32=res))
            row.append(f"{nm}k{splits} {ms*1e3:5.1f}")
    print(f"  M{M} N{N} K{K} res={int(res)}  " + " ".join(row))
force(None, 0)
