"""Where wall(pipeline.generate) goes on the host side (config 3): cProfile of one warm call + phase timers."""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sd_amd import pipeline, synth
dev = torch.device("cuda", 0)
models, _ = bench.build_models(dev, torch.bfloat16, False)
call = bench.GenerateCall(models, bench.CONFIGS[3], 0, 8, 50, dev)
call(); call(); torch.cuda.synchronize()
t0 = time.perf_counter(); call(); torch.cuda.synchronize(); print("warm call", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); call(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue())
