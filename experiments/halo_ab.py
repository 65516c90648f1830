"""Halo-staged conv3x3 (gemm4.hip) against the gathered form (gemm2.hip): results and time, rotated operands.
run: python experiments/halo_ab.py [quick]"""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()
OFF = 8 << 20
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1)
def t(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(n): fn(i)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / n)
    return sorted(ts)[2]
shapes = [(16, 64, 64, 320, 320), (16, 64, 64, 640, 320), (16, 64, 64, 960, 320), (16, 32, 32, 640, 640), (16, 32, 32, 1280, 640), (16, 32, 32, 1920, 640),
          (16, 16, 16, 1280, 1280), (16, 16, 16, 2560, 1280), (2, 64, 64, 320, 320), (2, 32, 32, 640, 640), (2, 16, 16, 1280, 1280), (1, 64, 64, 320, 320), (16, 8, 8, 1280, 1280), (16, 8, 8, 2560, 1280)]
if len(sys.argv) > 1: shapes = shapes[:1] + shapes[3:4] + shapes[6:7] + shapes[8:] if sys.argv[1] == 'q' else shapes[-2:]
for B, H, W, Cin, Cout in shapes:
    NR = max(2, min(8, int(600e6 / (B * H * W * Cin * 2))))
    xs = [rnd(B, H, W, Cin).bfloat16() for _ in range(NR)]
    w, b = (rnd(Cout, 9 * Cin) * ((9 * Cin) ** -0.5)).bfloat16(), torch.rand(Cout, device=dev, generator=g)
    res = rnd(B, H, W, Cout)
    for label, kw in (("bf16out", dict()), ("+res f32", dict(residual=res, out_f32=True)), ("f32 stats", dict(out_f32=True, want_stats=True))):
        outs = {}
        row = []
        for nm, impl in (("gather", OFF), ("halo", 2 << 20), ("hbw", 18 << 8)):
            L.sdamd_set_gemm_impl(impl)
            outs[nm] = ops.conv3x3(xs[0], w, b, **kw)
            ms = t(lambda i: ops.conv3x3(xs[i % NR], w, b, **kw), 2 * NR)
            row.append(f"{nm} {ms*1e3:6.1f}us {2.0*B*H*W*Cout*9*Cin/ms/1e9:5.0f}TF")
        d = (outs["halo"].float() - outs["gather"].float()).abs().max().item()
        dh = (outs["hbw"].float() - outs["halo"].float()).abs().max().item()
        ref = torch.nn.functional.conv2d(xs[0].double().permute(0, 3, 1, 2)[:1], w.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
        if "residual" in kw: ref = ref + res[:1].double()
        e = (outs["halo"][:1].double() - ref).abs().max().item()
        e0 = (outs["gather"][:1].double() - ref).abs().max().item()
        st = ""
        if "want_stats" in kw:
            sh, sg = ops._stats_of(outs["halo"]), ops._stats_of(outs["gather"])
            st = f" stats {'none' if sh is None else (sh.float() - sg.float()).abs().max().item() if sg is not None else 'halo only'}"
        print(f"conv B{B} {H}x{W} {Cin}->{Cout} {label:9s}: " + " | ".join(row) + f" | max|halo-gather| {d:.2e} |hbw-halo| {dh:.1e} err vs f64 halo {e:.2e} gather {e0:.2e}{st}", flush=True)
L.sdamd_set_gemm_impl(0)
