import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
B, HW, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.bfloat16 if len(sys.argv) > 4 and sys.argv[4] == "bf16" else torch.float32
x = torch.randn(B, HW, C, device="cuda").to(dt)
g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
for _ in range(20): ops.groupnorm(x, g, b, 1e-5, True, torch.bfloat16)
torch.cuda.synchronize()
