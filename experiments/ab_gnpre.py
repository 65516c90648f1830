"""GroupNorm with the producer's column statistics vs its own statistics pass, per UNet shape at batch 16, and the
cost the statistics add to the producing conv / GEMM epilogue."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, rounds=7, inner=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / inner)
    return sorted(ts)[len(ts) // 2]
for B, HW, C, dt in [(16, 4096, 320, torch.float32), (16, 1024, 640, torch.float32), (16, 256, 1280, torch.float32), (16, 4096, 320, torch.bfloat16), (16, 1024, 640, torch.bfloat16), (2, 4096, 320, torch.float32)]:
    x = torch.rand(B, HW, C, device=dev, generator=g).to(dt)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    st = torch.rand(B * HW // 64, C, 2, device=dev, generator=g)
    plain = t(lambda: ops.groupnorm(x, gamma, beta, 1e-5, True, out_dtype=torch.bfloat16))
    x._sd_colstats = st
    pre = t(lambda: ops.groupnorm(x, gamma, beta, 1e-5, True, out_dtype=torch.bfloat16))
    print(f"GN B{B} HW{HW} C{C} {dt}: own stats {plain*1e3:6.1f} us   producer stats {pre*1e3:6.1f} us")
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
for B, H, Cin, Cout, res in [(16, 64, 320, 320, True), (16, 64, 320, 320, False), (16, 32, 640, 640, True), (16, 32, 640, 640, False)]:
    x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), torch.rand(Cout, device=dev, generator=g)
    r = torch.rand(B, H, H, Cout, device=dev, generator=g) if res else None
    a = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False, out_f32=res))
    c = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False, out_f32=res, want_stats=True))
    print(f"conv B{B} H{H} {Cin}->{Cout} res={int(res)}: {a*1e3:6.1f} us   with stats {c*1e3:6.1f} us")
for M, N, K in [(65536, 320, 640), (16384, 640, 1280)]:
    a_, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = torch.rand(M, N, device=dev, generator=g)
    a = t(lambda: ops.gemm(a_, w, None, r, out_f32=True))
    c = t(lambda: ops.gemm(a_, w, None, r, out_f32=True, want_stats=True))
    print(f"gemm M{M} N{N} K{K} +res: {a*1e3:6.1f} us   with stats {c*1e3:6.1f} us")
