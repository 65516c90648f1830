"""Forced split-K on the 8x8 / 16x16 convs at batch 16 (debug = 32 + 8*tile + splits), in-process A/B."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"; L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
IMPLS = {"auto": 0, "s2": (32 + 2) << 8, "s3": (32 + 3) << 8, "s4": (32 + 4) << 8, "s5": (32 + 5) << 8, "s6": (32 + 6) << 8, "s7": (32 + 7) << 8}
for B, H, Cin, Cout in [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 640, 1280)]:
    x, w = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5)
    b = torch.rand(Cout, device=dev, generator=g)
    r = torch.rand(B, H, H, Cout, device=dev, generator=g)
    res = {k: [] for k in IMPLS}
    for rep in range(6):
        for k, v in IMPLS.items():
            L.sdamd_set_gemm_impl(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8): ops.conv3x3(x, w, b, r, 1, 0, False, out_f32=True)
            e.record(); torch.cuda.synchronize()
            if rep: res[k].append(s.elapsed_time(e) / 8)
    print(f"B{B} {H}x{H} {Cin}->{Cout}: " + "  ".join(f"{k} {sorted(v)[len(v)//2]*1e3:6.1f}us" for k, v in res.items()), flush=True)
