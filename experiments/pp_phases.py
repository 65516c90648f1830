"""Where a 256x256 ping-pong (GeGLU) tile's time goes: per-workgroup s_memtime stamps at kernel entry / first MFMA /
end of the K loop / end of the epilogue, from the -DSDAMD_PP_TIMING build (experiments/build_pp_timing.sh).
usage: SDAMD_LIB=experiments/micro/libsdamd_pptiming.so python experiments/pp_phases.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
L = _lib.lib()
L.sdamd_debug_pp_times.restype = ctypes.c_int
L.sdamd_debug_pp_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = "cuda"
for (M, N, K) in ((65536, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
    f = lambda: ops.gemm(a, w, b, None, ops.ACT_GEGLU)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    ntile = (M // 256) * (N // 256)
    buf = np.zeros(4 * 8192, dtype=np.int64)
    rc = L.sdamd_debug_pp_times(buf.ctypes.data, buf.size); assert rc == 0
    t = buf.reshape(-1, 4)[:ntile].astype(np.float64)
    xcd = np.arange(ntile) % 8                                 # the counters of the 8 XCDs are not synchronised
    for x in range(8): t[xcd == x] -= t[xcd == x, 0].min()
    us = s.elapsed_time(e) * 1e3
    clk = np.mean([t[xcd == x, 3].max() for x in range(8)]) / us    # ticks per us (a launch spans the event time on every XCD)
    d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]], 1) / clk
    order = np.argsort(t[:, 0]); start = t[order, 0] / clk
    print(f"M{M} N{N} K{K}: {ntile} tiles, launch {us:.1f} us, {clk:.0f} ticks/us")
    print(f"   per workgroup (us, mean / p10 / p90): entry->first MFMA {d[:,0].mean():.2f} / {np.percentile(d[:,0],10):.2f} / {np.percentile(d[:,0],90):.2f}"
          f"   K loop {d[:,1].mean():.2f} / {np.percentile(d[:,1],10):.2f} / {np.percentile(d[:,1],90):.2f}"
          f"   epilogue {d[:,2].mean():.2f} / {np.percentile(d[:,2],10):.2f} / {np.percentile(d[:,2],90):.2f}   whole {((t[:,3]-t[:,0])/clk).mean():.2f}")
    r = [start[i:i + 256].mean() for i in range(0, ntile, 256)]
    print("   mean entry time of successive groups of 256 workgroups (us):", " ".join(f"{x:.1f}" for x in r[:12]))
    # per XCD: the 32 CUs run one workgroup each; gap between the k-th end and the (k+32)-th entry
    gaps = []
    for x in range(8):
        st = np.sort(t[xcd == x, 0]) / clk; en = np.sort(t[xcd == x, 3]) / clk
        if len(st) > 32: gaps.append(st[32:] - en[:len(st) - 32])
    if gaps:
        gap = np.concatenate(gaps)
        print(f"   per XCD: (k+32)-th entry minus k-th end (us): mean {gap.mean():.2f}  p10 {np.percentile(gap,10):.2f}  p90 {np.percentile(gap,90):.2f}")
