"""One conv shape, a few launches, for rocprofv3 --pmc runs: pp_one.py <pp_bits> [dbg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
pp = int(sys.argv[1]); dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B, H, Cin, Cout = 8, 64, 512, 512
x = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).bfloat16()
b = torch.randn(Cout, device="cuda")
_lib.lib().sdamd_set_gemm_impl((pp << 20) | (dbg << 8))
for _ in range(5):
    ops.conv3x3(x, w, b)
torch.cuda.synchronize()
