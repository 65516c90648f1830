"""fp32 small-grid GEMM / conv shapes of UNet batch 2 (config 2): auto plan vs forced LDS-DMA tiles with deeper rings
(128x160 3-stage = idx 5, 64x160 2-stage = idx 6, 64x160 4-stage = idx 7) and K splits.  Are these launches bound by
the latency of one K-tile in flight?"""
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
dt = torch.float32
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(dt)
VARS = ((0, "128s2"), (5, "128s3"), (6, "64s2"), (7, "64s4"))
print("gemm f32")
for M, N, K, res in [(8192, 320, 320, True), (8192, 320, 320, False), (2048, 640, 640, True), (2048, 640, 640, False), (512, 1280, 1280, True),
                     (512, 3840, 1280, False), (8192, 320, 1280, True), (2048, 640, 2560, True), (512, 1280, 5120, True), (2048, 1920, 640, False),
                     (8192, 960, 320, False), (8192, 2560, 320, False), (512, 10240, 1280, False)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = torch.rand(M, N, device=dev, generator=g) if res else None
    force(None, 0)
    ref = ops.gemm(a, w, None, r)
    ms = t(lambda: ops.gemm(a, w, None, r))
    row = [f"auto {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:4.0f}TF |"]
    for idx, nm in VARS:
        for splits in (1, 2, 4):
            if splits > 1 and K // 32 // splits < 4: continue
            force(idx, splits)
            out = ops.gemm(a, w, None, r)
            err = float((out - ref).abs().max())
            ms = t(lambda: ops.gemm(a, w, None, r))
            row.append(f"{nm}k{splits} {ms*1e3:5.1f}" + ("" if err < 1e-3 else f"(ERR {err:.1e})"))
    print(f"  M{M} N{N} K{K} res={int(res)}  " + " ".join(row), flush=True)
print("conv f32")
for B, H, Cin, Cout, res in [(2, 8, 1280, 1280, True), (2, 8, 2560, 1280, False), (2, 16, 1280, 1280, True), (2, 32, 640, 640, True), (2, 64, 320, 320, True)]:
    x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), torch.rand(Cout, device=dev, generator=g)
    r = torch.rand(B, H, H, Cout, device=dev, generator=g) if res else None
    force(None, 0)
    ref = ops.conv3x3(x, w, b, r, 1, 0, False)
    ms = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False))
    row = [f"auto {ms*1e3:6.1f}us {2.0*B*H*H*Cout*9*Cin/ms/1e9:4.0f}TF |"]
    for idx, nm in VARS:
        for splits in (1, 2, 4, 7):
            force(idx, splits)
            out = ops.conv3x3(x, w, b, r, 1, 0, False)
            err = float((out - ref).abs().max())
            ms = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False))
            row.append(f"{nm}k{splits} {ms*1e3:5.1f}" + ("" if err < 1e-3 else f"(ERR {err:.1e})"))
    print(f"  B{B} H{H} {Cin}->{Cout} res={int(res)}  " + " ".join(row), flush=True)
force(None, 0)
