"""Per-launch-shape time table of one eager UNet step (+CFG/DDPM) at the bench batch (experiment)."""
import sys; sys.path.insert(0, '.')
import torch
import bench
from sd_amd import ops, engine
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
import os
if os.environ.get("SDAMD_IMPL"):   # library switch word for an A/B (sdamd_set_gemm_impl), e.g. 8388608 = halo-staged convs off
    from sd_amd import _lib
    _lib.lib().sdamd_set_gemm_impl(int(os.environ["SDAMD_IMPL"]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
models, _ = bench.build_models(dev, dtype, False)
job = bench.Txt2ImgJob(models, 0, B, 50, 7.5, dev, use_graph=False)
job.loop.reset(job.latents0); job.loop.step(); job.loop.step(); torch.cuda.synchronize()
ops.PROFILER = ops.LaunchTimer()
N = 3
for _ in range(N): job.loop.step()
rows = ops.PROFILER.by_shape(); ops.PROFILER = None
tot = sum(v["ms"] for v in rows.values()) / N
print(f"UNet batch {2*B} {dtype}: sum of timed launches {tot:.2f} ms/step")
acc = 0
for (kind, shape), v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
    ms = v["ms"] / N; acc += ms
    mf = kind.split("_")[0] in ("conv3x3", "gemm", "attention")
    rate = v["work"] / N / (ms * 1e-3) / (1e12 if mf else 1e9)
    print(f"{ms:7.3f} ms {100*ms/tot:5.1f}% cum {100*acc/tot:5.1f}%  x{v['launches']//N:3d}  {rate:8.1f} {'TF' if mf else 'GB/s'}  {kind:24s} {shape}")
