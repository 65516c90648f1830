"""fp32 attention: exact f32 MFMA kernel vs the bf16x3 kernel (sdamd_set_f32_mode): time and error against float64."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()
def t(fn, rounds=7, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / inner)
    return sorted(ts)[len(ts) // 2]
for B, H, d, Sq, Skv, causal, amp in [(2, 8, 40, 4096, 4096, False, 1.0), (2, 8, 40, 4096, 4096, False, 4.0), (2, 8, 80, 1024, 1024, False, 1.0), (2, 8, 40, 4096, 77, False, 1.0),
                                      (2, 8, 80, 1024, 77, False, 1.0), (2, 12, 64, 77, 77, True, 1.0), (16, 8, 40, 4096, 4096, False, 1.0)]:
    C = H * d
    q = torch.randn(B, Sq, C, device=dev, generator=g) * amp
    k = torch.randn(B, Skv, C, device=dev, generator=g)
    v = torch.randn(B, Skv, C, device=dev, generator=g)
    hd = lambda x, S: x.double().view(B, S, H, d).transpose(1, 2)
    if B * H * Sq * Skv <= 2 * 8 * 4096 * 4096:
        ref = torch.nn.functional.scaled_dot_product_attention(hd(q, Sq), hd(k, Skv), hd(v, Skv), is_causal=causal).transpose(1, 2).reshape(B, Sq, C)
    else:
        ref = None
    row = []
    for mode in ("exact", "bf16x3"):
        ops.set_f32_mode(mode)
        out = ops.attention(q, k, v, H, causal=causal)
        ms = t(lambda: ops.attention(q, k, v, H, causal=causal))
        err = float((out.double() - ref).norm() / ref.norm()) if ref is not None else float("nan")
        row.append(f"{mode} {ms*1e3:7.1f}us {4.0*B*H*Sq*Skv*d/ms/1e9:6.1f}TF err {err:.1e}")
    print(f"B{B} H{H} d{d} Sq{Sq} Skv{Skv} causal={int(causal)} amp={amp}: " + " | ".join(row), flush=True)
ops.set_f32_mode("bf16x3")
