import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sd_amd
from sd_amd import ops
torch.set_grad_enabled(False)
g = torch.Generator(device="cuda").manual_seed(0)
B, H, d, S = 16, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
qkv = torch.randn(B, S, 3 * H * d, device="cuda", generator=g).bfloat16()
C = H * d
if len(sys.argv) > 3:   # library debug word (sdamd_set_gemm_impl(debug << 8)): kernel variants / timing ablations
    from sd_amd import _lib
    _lib.lib().sdamd_set_gemm_impl(int(sys.argv[3]) << 8)
Skv = int(sys.argv[4]) if len(sys.argv) > 4 else S     # fewer keys than queries: cross attention (K|V from their own tensor)
kv = qkv[:, :Skv, C:].contiguous()
q = qkv[..., :C].contiguous() if len(sys.argv) > 4 else qkv[..., :C]   # (the cross attention's Q is a projection of its own)
for _ in range(3): ops.attention(q, kv[..., :C], kv[..., C:], H)
torch.cuda.synchronize()
