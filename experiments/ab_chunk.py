"""Does running the 64x64-level attention block on sub-batches (working set inside the 256 MB Infinity Cache) beat one
pass over the whole UNet batch?  Times UNET_AttentionBlock.run at UNet batch 16 whole vs in chunks (no concat)."""
import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import diffusion, engine, ops, synth
torch.set_grad_enabled(False)
dev = "cuda"
def t(fn, rounds=7, inner=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / inner)
    return sorted(ts)[len(ts) // 2]
for heads, dh, hw in ((8, 40, 64), (8, 80, 32)):
    C = heads * dh
    blk = engine.set_compute_dtype(synth.load_hashed(diffusion.UNET_AttentionBlock(heads, dh), "blk.attn.").to(dev), torch.bfloat16)
    res = engine.set_compute_dtype(synth.load_hashed(diffusion.UNET_ResidualBlock(C, C), "blk.res2.").to(dev), torch.bfloat16)
    B = 16
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, hw, hw, C, device=dev, generator=g)
    ctx = torch.randn(B, 77, 768, device=dev, generator=g)
    kv = blk.prompt_kv(ctx)
    tb = torch.randn(C, device=dev, generator=g)
    for nchunk in (1, 2, 4, 8):
        cb = B // nchunk
        xs = [x[i * cb:(i + 1) * cb].contiguous() for i in range(nchunk)]
        kvs = [kv[i * cb:(i + 1) * cb].contiguous() for i in range(nchunk)]
        ms_a = t(lambda: [blk.run(xs[i], kvs[i]) for i in range(nchunk)])
        ms_r = t(lambda: [res.run(xs[i], tb) for i in range(nchunk)])
        print(f"level {hw}x{hw} C{C}: {nchunk} chunk(s) of {cb}: attention block {ms_a:.3f} ms   residual block {ms_r:.3f} ms")
