"""First / last conv of the UNet and VAE (cin_small / cout_small) timing, A/B via SDAMD_LIB."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
dev = "cuda"
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 5)
    return sorted(ts)[2] * 1e3
out = []
for B, H, Cin, Cout, dt in [(16, 64, 320, 4, torch.float32), (8, 512, 128, 3, torch.bfloat16)]:
    V = 4 if dt == torch.float32 else 8
    x = torch.randn(B, H, H, Cin, device=dev).to(dt)
    w = torch.randn(9, Cin // V, Cout, V, device=dev).to(dt)
    b = torch.randn(Cout, device=dev)
    out.append(f"cout_small B{B} {H}x{H} {Cin}->{Cout} {str(dt)[6:]}: {t(lambda: ops.conv3x3_cout_small(x, w, b, Cout)):7.1f}us")
print("  ".join(out))
