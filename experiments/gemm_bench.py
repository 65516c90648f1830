"""A/B of the GEMM/conv kernel families on the UNet's bf16 problem sizes (batch 16), random data,
interleaved rounds in one process (experiment)."""
import sys; sys.path.insert(0, '.')
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
L = _lib.lib()
IMPLS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,0").split(",")]
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
gemms = [(65536, 2560, 320, 0, 0), (16384, 5120, 640, 0, 0), (4096, 10240, 1280, 0, 0), (65536, 320, 1280, 1, 1), (65536, 320, 320, 1, 1),
         (16384, 640, 2560, 1, 1), (16384, 640, 640, 1, 1), (4096, 1280, 5120, 1, 1), (65536, 320, 640, 1, 1), (65536, 960, 320, 0, 0),
         (4096, 1280, 1280, 1, 1), (16384, 1920, 640, 0, 0), (4096, 3840, 1280, 0, 0), (1024, 1280, 5120, 0, 1)]
convs = [(16, 64, 64, 320, 320, 1, 0, 1, 1), (16, 64, 64, 640, 320, 1, 0, 0, 0), (16, 32, 32, 640, 640, 1, 0, 1, 1), (16, 16, 16, 1280, 1280, 1, 0, 1, 1),
         (16, 8, 8, 1280, 1280, 1, 0, 1, 1), (16, 8, 8, 2560, 1280, 1, 0, 0, 0), (16, 16, 16, 2560, 1280, 1, 0, 0, 0), (16, 32, 32, 1280, 640, 1, 1, 0, 1),
         (16, 16, 16, 2560, 1280, 1, 1, 0, 1), (16, 64, 64, 960, 320, 1, 0, 0, 0), (16, 32, 32, 1920, 640, 1, 0, 0, 0), (16, 64, 64, 640, 320, 2, 0, 0, 1)]
def timeit(fn, rounds=5, inner=4):
    best = {}
    for i in IMPLS:
        L.sdamd_set_gemm_impl(i); fn(); 
    torch.cuda.synchronize()
    for r in range(rounds):
        for i in IMPLS:
            L.sdamd_set_gemm_impl(i)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(inner): fn()
            e.record(); torch.cuda.synchronize()
            best.setdefault(i, []).append(s.elapsed_time(e) / inner)
    return {i: sorted(v)[len(v) // 2] for i, v in best.items()}
tot = {i: 0.0 for i in IMPLS}
for M, N, K, res, f32 in gemms:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = (torch.rand(M, N, device=dev, generator=g) if f32 else rnd(M, N)) if res else None
    t = timeit(lambda: ops.gemm(a, w, None, r, out_f32=bool(f32)))
    fl = 2.0 * M * N * K
    for i in IMPLS: tot[i] += t[i]
    print(f"gemm M{M:6d} N{N:5d} K{K:5d} res{res} f32{f32}: " + "  ".join(f"impl{i} {t[i]*1e3:7.1f}us {fl/t[i]/1e9:7.1f}TF" for i in IMPLS), flush=True)
for B, H, W, Cin, Cout, stride, ups, res, f32 in convs:
    x, w = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5)
    Ho, Wo = (H * (2 if ups else 1)) // stride, (W * (2 if ups else 1)) // stride
    r = (torch.rand(B, Ho, Wo, Cout, device=dev, generator=g) if f32 else rnd(B, Ho, Wo, Cout)) if res else None
    bias = torch.rand(Cout, device=dev, generator=g)
    t = timeit(lambda: ops.conv3x3(x, w, bias, r, stride, 0, bool(ups), out_f32=bool(f32)))
    fl = 2.0 * B * Ho * Wo * Cout * 9 * Cin
    for i in IMPLS: tot[i] += t[i]
    print(f"conv B{B} {H}x{W} Cin{Cin:5d} Cout{Cout:5d} s{stride} up{ups} res{res} f32{f32}: " + "  ".join(f"impl{i} {t[i]*1e3:7.1f}us {fl/t[i]/1e9:7.1f}TF" for i in IMPLS), flush=True)
print("sum ms:", {i: round(v, 3) for i, v in tot.items()})
