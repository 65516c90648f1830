"""VAE decode of 8 latents at 512 x 512 (the tail of one config-3 generate() call): median time; for A/B runs of two
library builds (SDAMD_LIB=...)."""
import sys; sys.path.insert(0, '.')
import os, torch, sd_amd
from sd_amd import decoder, engine, ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda")
dm = engine.set_compute_dtype(synth.load_hashed(decoder.VAE_Decoder(), "decoder.").to(dev), torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(0)
lat = torch.randn(8, 4, 64, 64, device=dev, generator=g)
for _ in range(2): out = dm(lat)
torch.cuda.synchronize(); ts = []
for r in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); out = dm(lat); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
ts.sort()
print(f"{os.environ.get('SDAMD_LIB', 'in-tree')}: VAE decode of 8 images median {ts[3]:.2f} ms min {ts[0]:.2f}  |out| {float(out.float().abs().sum()):.3f}")
