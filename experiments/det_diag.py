"""Which launches plan differently at batch 1 and batch 8 under ops.plan_samples (labels carry tile / split / kernel family)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sd_amd import ops, pipeline, synth, engine
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16
models, _ = bench.build_models(dev, dt, False)
kw = dict(models=models, device="cuda", sampler="ddpm", steps=1, cfg_scale=7.5, tokenizer=synth.SyntheticTokenizer(), compute_dtype=dt,
          return_latents=True, use_graph=False, deterministic_plan=True)
def run(prompts, seed):
    ops.PROFILER = ops.LaunchTimer()
    img, lat = pipeline.generate(prompts, "", seed=seed, **kw)
    recs = [(r[0], r[4]) for r in ops.PROFILER.records]
    ops.PROFILER = None
    return recs, lat
r8, l8 = run([f"synthetic:{i}" for i in range(8)], 42)
r1, l1 = run("synthetic:0", 42)
print(len(r8), len(r1), "latents equal:", torch.equal(l8[0], l1[0]))
import re
for i, (a, b) in enumerate(zip(r8, r1)):
    if a[0] != b[0]:
        print(i, a, b)

# ---- drill down: fingerprints of sample 0 after every engine-level op
import hashlib
from sd_amd import diffusion, clip, decoder
trace = []
def fp(t, B):
    if not torch.is_tensor(t): return None
    x = t[0] if t.dim() >= 3 else t[: t.shape[0] // B]
    return hashlib.md5(x.contiguous().cpu().numpy().tobytes()).hexdigest()[:10]
def wrap(mod, name, B_of):
    orig = getattr(mod, name)
    def f(*a, **k):
        out = orig(*a, **k)
        o = out[0] if isinstance(out, tuple) else out
        trace.append((name, tuple(o.shape), fp(o, CUR_B)))
        return out
    setattr(mod, name, f)
for name in ("groupnorm", "conv3x3", "conv1x1", "conv1x1_presplit", "linear", "linear_geglu", "linear_hilo", "layernorm", "linear_ln", "conv1x1_ln",
             "conv3x3_trunk", "cross_kv"):
    wrap(engine, name, None)
for name in ("attention", "split_hilo", "conv3x3_cin_small", "conv3x3_cout_small", "cfg_ddpm_step"):
    wrap(ops, name, None)
CUR_B = 16
trace.clear(); run([f"synthetic:{i}" for i in range(8)], 42); t8 = list(trace)
CUR_B = 2
trace.clear(); run("synthetic:0", 42); t1 = list(trace)
print(len(t8), len(t1))
n = 0
for i, (a, b) in enumerate(zip(t8, t1)):
    if a[0] != b[0] or a[2] != b[2]:
        print(i, a, b); n += 1
        if n > 6: break
