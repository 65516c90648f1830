"""Forced-tile A/B on the 64x64- and 32x32-level convs / GEMMs at UNet batch 16: 128x160 vs 128x320 vs 256x320."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
L = _lib.lib()
TILES = {"128x160": 0, "128x320": 1, "256x256": 2, "256x320": 4}
def force(name, splits=1):
    L.sdamd_set_gemm_impl(0 if name is None else (32 + 8 * TILES[name] + splits) << 8)
def t(fn, rounds=5, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / inner)
    return sorted(ts)[len(ts) // 2]
print("conv3x3 (B,H,Cin,Cout,res)")
for B, H, Cin, Cout in [(16, 64, 320, 320), (16, 64, 640, 320), (16, 64, 960, 320), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 32, 320, 640)]:
    x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), torch.rand(Cout, device=dev, generator=g)
    res = torch.rand(B, H, H, Cout, device=dev, generator=g)
    for use_res in (False, True):
        row = []
        for name in ("128x160", "128x320", "256x320"):
            force(name)
            ms = t(lambda: ops.conv3x3(x, w, b, res if use_res else None, 1, 0, False, out_f32=use_res))
            row.append(f"{name} {ms*1e3:7.1f}us {2.0*B*H*H*Cout*9*Cin/ms/1e9:5.0f}TF")
        print(f"  H{H} {Cin:4d}->{Cout:4d} res={int(use_res)}  " + "   ".join(row))
print("gemm (M,N,K)")
for M, N, K in [(65536, 320, 320), (65536, 960, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 1920, 640), (16384, 640, 2560), (4096, 1280, 5120)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    row = []
    for name in ("128x160", "128x320", "256x320"):
        force(name)
        ms = t(lambda: ops.gemm(a, w, None, None))
        row.append(f"{name} {ms*1e3:7.1f}us {2.0*M*N*K/ms/1e9:5.0f}TF")
    print(f"  M{M} N{N} K{K}  " + "   ".join(row))
force(None)
