"""N eager UNet steps (+CFG/DDPM) at the bench batch: the command profiled for per-kernel PMC traffic."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
models, _ = bench.build_models(dev, torch.bfloat16, False)
job = bench.Txt2ImgJob(models, 0, 8, 50, 7.5, dev, use_graph=False)
job.loop.reset(job.latents0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3): job.loop.step()
torch.cuda.synchronize()
