"""Forced split-K on mid-size GEMMs (debug = 32 + 8*tile + splits), in-process A/B."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"; L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
IMPLS = {"auto": 0, "160s1": (32 + 1) << 8, "160s2": (32 + 2) << 8, "160s3": (32 + 3) << 8, "320s1": (32 + 8 + 1) << 8, "320s2": (32 + 8 + 2) << 8}
for M, N, K in [(4096, 1280, 1280), (4096, 1280, 2560), (4096, 1280, 5120), (16384, 640, 640), (16384, 640, 1280), (1024, 1280, 1280), (4096, 3840, 1280)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = torch.rand(M, N, device=dev, generator=g)
    res = {k: [] for k in IMPLS}
    for rep in range(6):
        for k, v in IMPLS.items():
            L.sdamd_set_gemm_impl(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8): ops.gemm(a, w, None, r, out_f32=True)
            e.record(); torch.cuda.synchronize()
            if rep: res[k].append(s.elapsed_time(e) / 8)
    print(f"M{M} N{N} K{K}: " + "  ".join(f"{k} {sorted(v)[len(v)//2]*1e3:6.1f}us" for k, v in res.items()), flush=True)
