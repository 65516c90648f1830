"""Where a GEMM / conv workgroup's time goes: per-workgroup s_memtime stamps at kernel entry / (ping-pong loop: first MFMA) /
end of the K loop / end of the epilogue, from the -DSDAMD_PP_TIMING build (experiments/build_pp_timing.sh), which writes them
into the split-K workspace of un-split launches.
usage: SDAMD_LIB=$PWD/experiments/micro/libsdamd_pptiming.so python experiments/pp_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda")
bf = torch.bfloat16
def rn(*shape, s=1.0): return (torch.randn(*shape, device=dev) * s)
cases = []
def gemm_case(name, M, N, K, res=False, f32=False, geglu=False, ln=False):
    a = rn(M, K).to(bf); w = rn(N, K, s=K ** -0.5).to(bf); b = rn(N); r = rn(M, N) if res else None
    if ln:
        g, be = rn(N), rn(N)
        cases.append((name, lambda: ops.gemm_ln(a, w, b, r, g, be, 1e-5)))
    else:
        cases.append((name, lambda: ops.gemm(a, w, b, r, ops.ACT_GEGLU if geglu else ops.ACT_NONE, out_f32=f32 or res)))
def conv_case(name, B, HW, Cin, Cout, res=False, f32=False):
    x = rn(B, HW, HW, Cin).to(bf); w = rn(Cout, 9 * Cin, s=(9 * Cin) ** -0.5).to(bf); b = rn(Cout)
    r = rn(B, HW, HW, Cout) if res else None
    cases.append((name, lambda: ops.conv3x3(x, w, b, r, out_f32=f32 or res)))
gemm_case("GeGLU 256x256 M65536 N2560 K320", 65536, 2560, 320, geglu=True)
gemm_case("GeGLU 256x256 M16384 N5120 K640", 16384, 5120, 640, geglu=True)
gemm_case("QKV 128x320 M65536 N960 K320 (bf16 out)", 65536, 960, 320)
gemm_case("out-proj + LN 128x320 M65536 N320 K320 +res", 65536, 320, 320, res=True, ln=True)
gemm_case("128x160 M16384 N640 K640 +res f32", 16384, 640, 640, res=True)
gemm_case("128x160 M16384 N1920 K640 (bf16 out)", 16384, 1920, 640)
gemm_case("128x160 M4096 N1280 K1280 +res f32", 4096, 1280, 1280, res=True)
conv_case("conv halo-128 B16 64x64 320->320 +res f32", 16, 64, 320, 320, res=True)
conv_case("conv halo-128 B16 64x64 640->320", 16, 64, 640, 320)
conv_case("conv halo-128 B16 32x32 640->640 +res f32", 16, 32, 640, 640, res=True)
conv_case("conv big-wave B16 32x32 1280->640", 16, 32, 1280, 640)
ws = ops._gemm_workspace(dev)
wsi = ws.view(torch.int64) if ws.dtype != torch.int64 else ws
for name, f in cases:
    for _ in range(2): f()
    torch.cuda.synchronize()
    wsi[:8 * 4096].zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    t = wsi[:8 * 4096].cpu().numpy().reshape(-1, 8).astype(np.float64)
    live = t[:, 0] > 0
    n = int(live.sum())
    if n == 0:
        print(f"{name}: launch {us:.1f} us -- no stamps (split launch?)"); continue
    t = t[live]
    has1 = (t[:, 1] > 0).all()
    whole = t[:, 3] - t[:, 0]
    pre = (t[:, 1] - t[:, 0]) if has1 else np.zeros(n)
    loop = t[:, 2] - (t[:, 1] if has1 else t[:, 0])
    epi = t[:, 3] - t[:, 2]
    # the XCDs' counters are not synchronised: only differences inside a workgroup are meaningful.  Scale: ticks of a
    # workgroup's lifetime x (workgroups / resident slots) ~ launch time; report shares and the implied microseconds
    sh = lambda x: 100.0 * x.mean() / whole.mean()
    print(f"{name}: {n} workgroups, launch {us:.1f} us; a workgroup's life: "
          + (f"before first MFMA {sh(pre):.0f} %, K loop {sh(loop):.0f} %" if has1 else f"prologue + K loop {sh(loop):.0f} %")
          + f", epilogue {sh(epi):.0f} %   (ticks: whole {whole.mean():.0f}, epilogue {epi.mean():.0f}, p90 {np.percentile(epi, 90):.0f})", flush=True)
    if (t[:, 4] > 0).all() and (t[:, 5] > 0).all() and (t[:, 6] > 0).all():
        print(f"      inside the epilogue (ticks, mean): bias loads + barrier {(t[:,4]-t[:,2]).mean():.0f}, first pass into LDS {(t[:,5]-t[:,4]).mean():.0f}, "
              f"first pass stored {(t[:,6]-t[:,5]).mean():.0f}, the rest {(t[:,3]-t[:,6]).mean():.0f}", flush=True)
