"""fp32 GEMM / conv on the bf16 matrix pipe (three-term operand split, six MFMA products; gemm2.hip split3_bf16) against
the exact-f32 MFMA kernels: time per launch and error of both against a float64 reference, config-2 (UNet batch 2) and
VAE shapes (sdamd_set_f32_mode)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()
def emu(on): L.sdamd_set_f32_mode(1 if on else 0)
def t(fn, rounds=7, inner=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / inner)
    return sorted(ts)[len(ts) // 2]
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
def rel(a, ref): return float((a.double() - ref).norm() / ref.norm())
print("gemm f32")
for M, N, K, res in [(8192, 320, 320, True), (2048, 640, 640, True), (512, 1280, 1280, True), (512, 3840, 1280, False), (8192, 320, 1280, True),
                     (2048, 640, 2560, True), (512, 1280, 5120, True), (2048, 1920, 640, False), (8192, 960, 320, False), (8192, 2560, 320, False),
                     (512, 10240, 1280, False), (2048, 5120, 640, False), (65536, 320, 320, True), (65536, 2560, 320, False)]:
    a, w = rnd(M, K), rnd(N, K) * (K ** -0.5)
    r = rnd(M, N) if res else None
    ref = a.double() @ w.double().t() + (r.double() if res else 0)
    row = []
    for on, pl in ((False, False), (True, False), (True, True)):
        emu(on); ops.USE_WEIGHT_PLANES = pl
        out = ops.gemm(a, w, None, r)
        if pl: assert torch.equal(out, prev), float((out - prev).abs().max())
        prev = out
        ms = t(lambda: ops.gemm(a, w, None, r))
        row.append(f"{('x3+planes' if pl else 'bf16x3') if on else 'exact '} {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:4.0f}TF err {rel(out, ref):.1e}")
    print(f"  M{M} N{N} K{K} res={int(res)}  " + " | ".join(row), flush=True)
print("conv f32")
for B, H, Cin, Cout, res in [(2, 8, 1280, 1280, True), (2, 16, 1280, 1280, True), (2, 32, 640, 640, True), (2, 64, 320, 320, True), (2, 64, 640, 320, False),
                             (2, 32, 1920, 640, False), (1, 128, 512, 512, False), (1, 256, 256, 256, False), (16, 64, 320, 320, True)]:
    x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5), rnd(Cout)
    r = rnd(B, H, H, Cout) if res else None
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
    if res: ref = ref + r.double()
    row = []
    for on, pl in ((False, False), (True, False), (True, True)):
        emu(on); ops.USE_WEIGHT_PLANES = pl
        out = ops.conv3x3(x, w, b, r, 1, 0, False)
        if pl: assert torch.equal(out, prev), float((out - prev).abs().max())
        prev = out
        ms = t(lambda: ops.conv3x3(x, w, b, r, 1, 0, False))
        row.append(f"{('x3+planes' if pl else 'bf16x3') if on else 'exact '} {ms*1e3:6.1f}us {2.0*B*H*H*Cout*9*Cin/ms/1e9:4.0f}TF err {rel(out, ref):.1e}")
    print(f"  B{B} H{H} {Cin}->{Cout} res={int(res)}  " + " | ".join(row), flush=True)
emu(True)
