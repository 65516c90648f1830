"""64x64-level GEMMs on the f32 trunk (cold operands: buffer sets > the Infinity Cache): forced tile / ring-depth variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
L = _lib.lib(); dev = "cuda"
VARIANTS = {"auto": 0, "128x160": 33, "128x160 3stage": 73, "128x320": 41, "64x160": 81, "64x160 4stage": 89, "256x320": 65}
def bench(M, N, K, res, hilo, nset=3, iters=9, rounds=3):
    sets = []
    for _ in range(nset):
        a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if res else None
        sets.append((a, w, b, r))
    call = lambda a, w, b, r: ops.gemm(a, w, b, r, out_hilo=hilo, out_f32=(res and not hilo))
    out = []
    for name, dbg in VARIANTS.items():
        L.sdamd_set_gemm_impl(dbg << 8)
        try:
            ts = []
            for _ in range(rounds):
                call(*sets[0]); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for i in range(iters): call(*sets[i % nset])
                e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / iters * 1e3)
            out.append(f"{name} {sorted(ts)[1]:.1f}")
        except Exception as ex:
            out.append(f"{name} n/a")
    L.sdamd_set_gemm_impl(0)
    print(f"M{M} N{N} K{K}{' +res' if res else ''}{' hilo' if hilo else ''}: " + " | ".join(out), flush=True)
bench(65536, 320, 1280, True, True)
bench(65536, 320, 640, True, False)
bench(65536, 960, 320, False, False)
bench(65536, 320, 320, False, False)
bench(16384, 640, 2560, True, True)
bench(16384, 1920, 640, False, False)
