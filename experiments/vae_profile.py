import sys; sys.path.insert(0, '.')
import torch, sd_amd
from sd_amd import decoder, engine, ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda")
dm = engine.set_compute_dtype(synth.load_hashed(decoder.VAE_Decoder(), "decoder.").to(dev), torch.bfloat16)
lat = torch.randn(8, 4, 64, 64, device=dev)
dm(lat); torch.cuda.synchronize()
ops.PROFILER = ops.LaunchTimer()
dm(lat)
rows = ops.PROFILER.by_shape(); ops.PROFILER = None
for (kind, shape), v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print(f"{v['ms']:7.3f} ms x{v['launches']:3d} {v['work']/v['ms']/1e9:8.1f} TF {kind:28s} {shape}")
