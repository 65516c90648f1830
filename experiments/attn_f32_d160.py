import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0); L=_lib.lib()
def t(fn, rounds=7, inner=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for r in range(rounds):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e)/inner)
    return sorted(ts)[len(ts)//2]
for B,H,d,Sq,Skv in [(2,8,160,256,256),(2,8,160,256,77),(16,8,160,256,256),(2,8,160,64,64)]:
    C=H*d
    q=torch.randn(B,Sq,C,device=dev,generator=g)*2; k=torch.randn(B,Skv,C,device=dev,generator=g); v=torch.randn(B,Skv,C,device=dev,generator=g)
    hd=lambda x,S: x.double().view(B,S,H,d).transpose(1,2)
    ref=torch.nn.functional.scaled_dot_product_attention(hd(q,Sq),hd(k,Skv),hd(v,Skv)).transpose(1,2).reshape(B,Sq,C)
    row=[]
    for dbg in (0,19):
        L.sdamd_set_gemm_impl(dbg<<8)
        out=ops.attention(q,k,v,H); ms=t(lambda: ops.attention(q,k,v,H))
        row.append(f"{'x3' if dbg else 'exact'} {ms*1e3:6.1f}us {4.0*B*H*Sq*Skv*d/ms/1e9:5.1f}TF err {float((out.double()-ref).norm()/ref.norm()):.1e}")
    print(f"B{B} S{Sq}x{Skv} d{d}: "+" | ".join(row))
L.sdamd_set_gemm_impl(0)
