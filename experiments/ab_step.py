"""Same-process A/B of one captured UNet step (bf16, UNet batch 2*B) under a Python-level switch.
usage: ab_step.py <module.attr | lib:debug> [batch] [f32]   e.g.  ab_step.py ops.USE_PRODUCER_STATS 8"""
import sys; sys.path.insert(0, '.')
import importlib, torch, sd_amd
from sd_amd import clip, diffusion, engine, ops, pipeline, synth
torch.set_grad_enabled(False)
# "lib:<debug>" arms: False = sdamd_set_gemm_impl(<debug> << 8) while capturing, True = the library's defaults
if sys.argv[1].startswith("lib:"):
    from sd_amd import _lib
    class _LibSwitch:
        def __setattr__(self, name, val):
            if name == "f32exact": _lib.lib().sdamd_set_f32_mode(1 if val else 0)   # False arm = exact f32 MFMA
            else: _lib.lib().sdamd_set_gemm_impl(0 if val else int(name) << 8)
    target, attr = _LibSwitch(), sys.argv[1][4:]
else:
    mod, attr = sys.argv[1].rsplit(".", 1)
    target = importlib.import_module("sd_amd." + mod)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
DT = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.bfloat16
dev = torch.device("cuda")
dm = engine.set_compute_dtype(synth.load_hashed(diffusion.Diffusion(), "diffusion.").to(dev), DT)
cm = synth.load_hashed(clip.CLIP(), "clip.").to(dev)
pos = torch.tensor([synth.synthetic_token_ids(i) for i in range(B)], device=dev)
neg = torch.tensor([synth.EMPTY_PROMPT_IDS] * B, device=dev)
ctx = pipeline.encode_prompts(cm, pos, neg)
ts, coef, noise, lat = pipeline.prepare_latents([42 + i for i in range(B)], pipeline.DDPMSampler, 50, (1, 4, 64, 64), dev)
loops = {}
for val in (False, True):
    setattr(target, attr, val)
    loops[val] = pipeline.DenoiseLoop(dm, ctx, ts, coef, noise, lat, 7.5)
setattr(target, attr, True)
def step_ms(loop, n=12):
    loop.reset(lat)
    loop.step(); loop.step()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): loop.step()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
res = {False: [], True: []}
for r in range(6):
    for val in (False, True):
        res[val].append(step_ms(loops[val]))
for val in (False, True):
    v = sorted(res[val])
    print(f"{sys.argv[1]}={val}: UNet step median {v[len(v)//2]:.3f} ms  min {v[0]:.3f}  (batch {2*B})")
a, b = sorted(res[False])[3], sorted(res[True])[3]
print(f"delta {b - a:+.3f} ms ({100 * (b - a) / a:+.2f} %)")
# same latents either way (to rounding)
loops[False].reset(lat); loops[True].reset(lat)
for _ in range(3): loops[False].step(); loops[True].step()
d = float((loops[False].latents - loops[True].latents).norm() / loops[False].latents.norm())
print(f"rel-L2 between the two arms after 3 steps: {d:.2e}")
