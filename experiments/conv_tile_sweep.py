"""Forced tile / split-K sweep over the UNet's conv3x3 shapes at UNet batch 16, HBM-cold (rotating buffer sets),
with and without the f32 residual epilogue: input for gemm2_choose_tile (experiment)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"; L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
TILES = ["160", "320"]
cfgs = [(0, 1), (0, 2), (0, 4), (1, 1), (1, 2), (1, 4)]
NSET = 4
def run(fn, fl, label):
    out = []
    L.sdamd_set_gemm_impl(0)
    auto = ops._tile_tag(*label[1]) if label[1] else ""
    res = {}
    for tile, s in [(-1, 0)] + cfgs:
        L.sdamd_set_gemm_impl(0 if tile < 0 else (32 + 8 * tile + s) << 8)
        for i in range(NSET): fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); [fn(i % NSET) for i in range(2 * NSET)]; b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / (2 * NSET))
        res["auto" if tile < 0 else f"{TILES[tile]}/S{s}"] = sorted(ts)[1]
    L.sdamd_set_gemm_impl(0)
    best = min((v, k) for k, v in res.items() if k != "auto")
    print(f"{label[0]:42s} auto({auto}) {res['auto']*1e3:6.1f}us {fl/res['auto']/1e9:5.0f}TF | best {best[1]} {best[0]*1e3:6.1f}us ({100*(res['auto']/best[0]-1):+.1f}%) | " +
          " ".join(f"{k} {v*1e3:.0f}" for k, v in res.items() if k != "auto"), flush=True)
B = 16
shapes = [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 320, 640), (32, 640, 640), (32, 960, 640), (32, 1280, 640), (32, 1920, 640),
          (16, 640, 1280), (16, 1280, 1280), (16, 1920, 1280), (16, 2560, 1280)]
for H, Cin, Cout in shapes:
    w = rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5)
    bias = torch.rand(Cout, device=dev, generator=g)
    X = [rnd(B, H, H, Cin) for _ in range(NSET)]
    R = [torch.rand(B, H, H, Cout, device=dev, generator=g) for _ in range(NSET)]
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    ws = ops._gemm_workspace(torch.device(dev, 0))
    run(lambda i: ops.conv3x3(X[i], w, bias, None), fl, (f"conv {H}x{H} {Cin}->{Cout} bf16out", (B * H * H, Cout, 9 * Cin, torch.bfloat16, ws)))
    if Cin == Cout or Cin in (320, 640, 1280):
        run(lambda i: ops.conv3x3(X[i], w, bias, R[i], out_f32=True), fl, (f"conv {H}x{H} {Cin}->{Cout} +res f32out", None))
    del X, R
