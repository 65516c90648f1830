"""LayerNorm timing on the UNet's shapes at batch 16 (A/B between library builds via SDAMD_LIB)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
out = []
for rows, C in [(65536, 320), (16384, 640), (4096, 1280), (8192, 320), (154, 768)]:
    x = torch.randn(rows, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    fn = lambda: ops.layernorm(x, g, b, 1e-5, torch.bfloat16)
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10)
    t = sorted(ts)[2]
    out.append(f"rows{rows} C{C}: {t*1e3:6.1f}us {rows*C*6/t/1e6:6.0f}GB/s")
print("  ".join(out))
