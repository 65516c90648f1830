"""A/B of the fused-GeGLU GEMM tile (256x256 8-wave vs 128x128 4-wave) on the UNet's shapes at batch 16."""
import sys; sys.path.insert(0, '.')
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
L = _lib.lib()
IMPLS = [0, 6 << 8]
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
for M, N, K in [(65536, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280)]:
    a, w = rnd(M, K), ops.geglu_pack(rnd(N, K) * (K ** -0.5))
    b = torch.rand(N, device=dev, generator=g)
    outs = {}
    res = {i: [] for i in IMPLS}
    for r in range(6):
        for i in IMPLS:
            L.sdamd_set_gemm_impl(i)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4): outs[i] = ops.gemm(a, w, b, None, ops.ACT_GEGLU)
            e.record(); torch.cuda.synchronize()
            if r: res[i].append(s.elapsed_time(e) / 4)
    assert torch.equal(outs[IMPLS[0]], outs[IMPLS[1]])
    print(f"geglu M{M} N{N} K{K}: " + "  ".join(f"impl{i} {sorted(v)[len(v)//2]*1e3:7.1f}us {2.0*M*N*K/sorted(v)[len(v)//2]/1e9:7.1f}TF" for i, v in res.items()), flush=True)
