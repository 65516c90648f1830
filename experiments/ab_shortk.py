"""Short-K GEMMs with an f32 residual / output (attention out-proj, conv_input, conv_output): forced tile A/B
(debug = 32 + 8*tile + splits) and the bytes they move.  M is for UNet batch 16."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"; L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
IMPLS = {"auto": 0, "160": (32 + 1) << 8, "320": (32 + 8 + 1) << 8, "256": (32 + 16 + 1) << 8}
for M, N, K, res, f32 in [(65536, 320, 320, 1, 1), (65536, 320, 640, 1, 1), (65536, 320, 1280, 1, 1), (65536, 320, 320, 0, 1), (65536, 320, 320, 0, 0),
                          (16384, 640, 640, 1, 1), (16384, 640, 1280, 1, 1), (16384, 640, 640, 0, 1), (4096, 1280, 1280, 1, 1), (65536, 960, 320, 0, 0)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = torch.rand(M, N, device=dev, generator=g) if res else None
    out = {k: [] for k in IMPLS}
    for rep in range(6):
        for k, v in IMPLS.items():
            if k == "256" and N % 256: continue
            L.sdamd_set_gemm_impl(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8): ops.gemm(a, w, None, r, out_f32=bool(f32))
            e.record(); torch.cuda.synchronize()
            if rep: out[k].append(s.elapsed_time(e) / 8)
    L.sdamd_set_gemm_impl(0)
    mb = (M * K * 2 + N * K * 2 + M * N * (4 if f32 else 2) + (M * N * 4 if res else 0)) / 1e6
    print(f"M{M} N{N} K{K} res{res} f32{f32} {mb:6.0f} MB: " + "  ".join(
        f"{k} {sorted(v)[len(v)//2]*1e3:6.1f}us ({mb / sorted(v)[len(v)//2] / 1e3:4.2f} TB/s)" for k, v in out.items() if v), flush=True)
