"""One VAE-sized conv on the 256x256 LDS-DMA tile (128 KB dynamic LDS): the launch at which a rocprofv3 --pmc pass over
bench.py aborted in round 1.  Run as:  rocprofv3 --pmc FETCH_SIZE -- python3 experiments/pmc_one_vae_conv.py"""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
B, H, C = 8, 64, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(B, H, H, C, device="cuda", generator=g) - 0.5).bfloat16()
w = ((torch.rand(C, 9 * C, device="cuda", generator=g) - 0.5) * (9 * C) ** -0.5).bfloat16()
b = torch.zeros(C, device="cuda")
t = _lib.lib().sdamd_gemm_tile(B * H * H, C, 9 * C, 1, 1, ops.GEMM_WS_BYTES, 0)
print("tile plan", t, flush=True)
for i in range(3):
    y = ops.conv3x3(x, w, b)
    torch.cuda.synchronize()
    print("launch", i, "ok", float(y.float().abs().mean()), flush=True)
