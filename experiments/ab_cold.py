"""HBM-cold timing of the f32-epilogue GEMMs / convs: every call works on a different buffer set and the sets
together exceed the 256 MB Infinity Cache, as inside a UNet step.  Run once per library build (SDAMD_LIB=...)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
NSET = 6
def timeit(fn):
    for i in range(NSET): fn(i)
    best = []
    for rep in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(NSET * 2): fn(i % NSET)
        e.record(); torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / (NSET * 2))
    return sorted(best)[len(best) // 2] * 1e3
for M, N, K, res, f32 in [(65536, 320, 320, 1, 1), (65536, 320, 640, 1, 1), (65536, 320, 1280, 1, 1), (65536, 320, 320, 0, 1),
                          (16384, 640, 640, 1, 1), (16384, 640, 1280, 1, 1), (4096, 1280, 1280, 1, 1), (16384, 640, 2560, 1, 1)]:
    w = rnd(N, K) * (K ** -0.5)
    A = [rnd(M, K) for _ in range(NSET)]
    R = [torch.rand(M, N, device=dev, generator=g) for _ in range(NSET)] if res else [None] * NSET
    us = timeit(lambda i: ops.gemm(A[i], w, None, R[i], out_f32=bool(f32)))
    mb = (M * K * 2 + N * K * 2 + M * N * (4 if f32 else 2) + (M * N * 4 if res else 0)) / 1e6
    print(f"gemm M{M} N{N} K{K} res{res} f32{f32}: {us:7.1f} us  {mb / us / 1e3:4.2f} TB/s  {2 * M * N * K / us / 1e6:6.0f} TF", flush=True)
    del A, R
for B, H, Cin, Cout, res in [(16, 64, 320, 320, 1), (16, 64, 320, 320, 0), (16, 32, 640, 640, 1), (16, 16, 1280, 1280, 1), (16, 32, 640, 640, 0)]:
    w = rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5)
    X = [rnd(B, H, H, Cin) for _ in range(NSET)]
    R = [torch.rand(B, H, H, Cout, device=dev, generator=g) for _ in range(NSET)] if res else [None] * NSET
    us = timeit(lambda i: ops.conv3x3(X[i], w, None, R[i], out_f32=bool(res)))
    print(f"conv B{B} H{H} Cin{Cin} Cout{Cout} res{res}: {us:7.1f} us  {2 * B * H * H * Cout * 9 * Cin / us / 1e6:6.0f} TF", flush=True)
    del X, R
