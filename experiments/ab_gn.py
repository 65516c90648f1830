"""GroupNorm timing on the UNet's shapes at batch 16 (A/B between library builds via SDAMD_LIB)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
out = []
for B, HW, C, dt in [(16, 4096, 320, torch.float32), (16, 4096, 320, torch.bfloat16), (16, 1024, 640, torch.float32), (16, 256, 1280, torch.float32), (16, 64, 1280, torch.float32), (2, 4096, 320, torch.float32)]:
    x = torch.randn(B, HW, C, device=dev).to(dt)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    fn = lambda: ops.groupnorm(x, g, b, 1e-5, True, torch.bfloat16)
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10)
    out.append(f"B{B} HW{HW} C{C} {str(dt)[6:]}: {sorted(ts)[2]*1e3:6.1f}us")
print("  ".join(out))
