import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
L = _lib.lib()
torch.set_grad_enabled(False)
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
def t(fn, rounds=7, inner=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for r in range(rounds):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e)/inner)
    return sorted(ts)[len(ts)//2]
for B,H,d,Sq in [(16,8,40,4096),(16,8,80,1024),(16,8,160,256),(2,8,40,4096),(2,8,80,1024),(2,8,160,256),(16,8,40,9216),(3,8,40,300)]:
    C=H*d
    q=torch.randn(B,Sq,C,device=dev,generator=g).bfloat16()
    row=[]
    for Skv in (32,64,77,96,128):
        kv=torch.randn(B,Skv,2*C,device=dev,generator=g).bfloat16()
        L.sdamd_set_gemm_impl(21 << 8); ref = ops.attention(q,kv[...,:C],kv[...,C:],H); ms0=t(lambda: ops.attention(q,kv[...,:C],kv[...,C:],H))
        L.sdamd_set_gemm_impl(0); out = ops.attention(q,kv[...,:C],kv[...,C:],H); ms=t(lambda: ops.attention(q,kv[...,:C],kv[...,C:],H))
        err = float((out.float()-ref.float()).norm()/ref.float().norm())
        row.append(f"Skv{Skv} {ms0*1e3:5.1f}->{ms*1e3:5.1f}us ({err:.0e})")
    print(f"B{B} Sq{Sq} d{d}: "+"  ".join(row), f" (Q+O bytes {2*B*Sq*C*2/1e6:.0f} MB)")
