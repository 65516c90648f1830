"""Latency-bound mid-size GEMMs (one workgroup per CU or fewer): forced tile / ring depth / split-K variants, rotated buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
L = _lib.lib()
dev = "cuda"
VARIANTS = {"auto": 0, "128x160 s1": 32 + 0 + 1, "128x160 s2": 32 + 0 + 2, "128x160 3stage": 32 + 40 + 1, "128x160 3stage s2": 32 + 40 + 2,
            "128x320": 32 + 8 + 1, "64x160": 32 + 48 + 1, "64x160 4stage": 32 + 56 + 1, "64x160 4stage s2": 32 + 56 + 2}
def bench(M, N, K, res, nset=4, iters=12, rounds=3):
    sets = []
    for _ in range(nset):
        a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if res else None
        sets.append((a, w, b, r))
    out = []
    for name, dbg in VARIANTS.items():
        L.sdamd_set_gemm_impl(dbg << 8)
        try:
            ts = []
            for _ in range(rounds):
                ops.gemm(*sets[0], out_f32=True); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for i in range(iters): ops.gemm(*sets[i % nset], out_f32=True)
                e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / iters * 1e3)
            out.append(f"{name} {sorted(ts)[1]:.1f}")
        except Exception as ex:
            out.append(f"{name} n/a")
    L.sdamd_set_gemm_impl(0)
    print(f"M{M} N{N} K{K}{' +res' if res else ''}: " + " | ".join(out), flush=True)
for shape in ((4096, 1280, 1280, True), (16384, 640, 640, True), (4096, 1280, 2560, True), (16384, 640, 1280, True), (4096, 3840, 1280, False),
              (16384, 1920, 640, False), (4096, 1280, 5120, True), (1024, 1280, 1280, True)):
    bench(*shape)
