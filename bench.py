#!/usr/bin/env python3
"""Headline benchmark: images/sec of pipeline.generate() @ 50-step DDPM + CFG (BASELINE.json), one process per GPU.

    python bench.py --gpus 1 --steps 2 --warmup 1                 (config 3, the single-GPU headline)
    python bench.py --config {3,4,5} --gpus N ...                 (plain python: spawns its own N ranks through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--config C]

One "step" = ONE CALL of pipeline.generate() on this rank's images, i.e. SURVEY.md §8d's wall(generate): tokenise ->
CLIP encode of 2B prompts -> host noise draw (the reference's CPU mt19937 stream) + upload -> (config 4: VAE encode +
add_noise) -> all denoising steps as ONE hipGraph launch [UNet forward on 2B latents + fused CFG/DDPM step, per step]
-> VAE decode -> uint8 -> (N>1) the one RCCL all-gather of the images -> device->host copy.  Weights are resident and
graph capture / weight repacking happen in warm-up.

  --config 3   txt2img 512x512, bf16, 8 images per GPU                         (weak scaling; default)
  --config 4   img2img strength 0.8 (intended semantics: last 40 of 50 steps), 512x512, bf16, GLOBAL batch 64 sharded
               over the N GPUs (64/N images per GPU)                           (strong scaling)
  --config 5   txt2img 768x768 (latent 96x96, 9216-token self-attention), bf16, 32 images per GPU  (weak scaling)

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the dominant kernel symbol of the timed workload, timed live with HIP events on the launch stream
                in one extra eager (un-graphed) pass of the same call; FLOPs are the REFERENCE op's,
  cpu_baseline  the CPU oracle (oracle/sd_oracle.py, a port) timed on this host's cores on a bounded sample of
                BASELINE config 1 (2 UNet steps + 1 VAE decode + 2 CLIP encodes), with and without the reference's
                per-step preview decode, CPU model and thread count stated,
  extra         the same workload with noise pre-drawn and resident (config 3), UNet step latencies (median of 20
                one-step graph replays after 3 warm-ups: bf16 at the bench batch, bf16 and fp32 at UNet batch 2 =
                config 2) and the per-kernel-family breakdown.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (importing torch / sd_amd does not initialise the GPU; main() spawns ranks before any GPU call)

import sd_amd  # noqa: E402,F401
from sd_amd import clip, decoder, diffusion, distributed, engine, ops, pipeline, synth  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3, "hbm": 8000.0}  # MI355X_MICROARCH.md: dense TFLOP/s, GB/s
PMC_TRAFFIC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")   # newest first
# BASELINE.json configs 3-5.  gflop = algorithmic GFLOP per image; batch = images per GPU (weak scaling)
# or global_batch = images of the whole job (strong scaling: global_batch / N per GPU).
CONFIGS = {
    3: dict(res=512, batch=8, global_batch=None, img2img=False, gflop=82866.0, scaling="weak",    # 50 x 1606.5 + 2514.5 + 26.6
            what="txt2img 512x512, 50-step DDPM, CFG 7.5, bf16, batch 8 per GPU"),
    4: dict(res=512, batch=None, global_batch=64, img2img=True, gflop=67918.0, scaling="strong",  # 40 x 1606.5 + 2514.5 + 1116.7 + 26.6
            what="img2img strength 0.8 (intended semantics: VAE encode + add_noise + the last 40 of 50 DDPM steps from t=780; the "
                 "reference's literal set_strength leaves no timesteps, SURVEY.md §8a S4), 512x512, CFG 7.5, bf16, global batch 64 "
                 "sharded over the GPUs"),
    5: dict(res=768, batch=32, global_batch=None, img2img=False, gflop=220591.0, scaling="weak",  # 50 x 4296.2 + 5754.3 + 26.6
            what="txt2img 768x768 (latent 96x96, 9216-token self-attention), 50-step DDPM, CFG 7.5, bf16, batch 32 per GPU"),
}


def kernel_symbol(kind):
    """Bench label -> substring of the kernel's demangled name in a rocprofv3 trace."""
    parts = kind.split("_")
    if parts[0] in ("gemm", "conv3x3") and len(parts) >= 3 and "x" in parts[2]:
        bm, bn = parts[2].split("x")
        conv = "true" if parts[0] == "conv3x3" else "false"
        t = "unsigned short" if parts[1] == "bf16" else "float"
        wg = {"128x320": "2, 4, 2", "128x160": "2, 2, 2", "64x160": "2, 2, 2", "256x256": "2, 4, 2", "256x320": "4, 2, 2", "256x128": "4, 2, 2"}.get(parts[2])
        if wg is not None:      # LDS-DMA family (gemm2.hip); last argument: 1 = residual-prefetch, 2 = GeGLU-only, 3 = fused LayerNorm epilogue
            epi = 1 if 'rp' in parts[3:] else 2 if 'geglu' in parts[3:] else 3 if 'ln' in parts[3:] else 0
            if 'hbwup' in parts[3:]:
                return "gemm2hbwup_kernel"                                            # ... its phase form (upsample + conv3x3)
            if 'hbw' in parts[3:]:
                return "gemm2hbw_kernel"                                              # 256-row big-wave halo conv3x3 (gemm5.hip)
            if 'halo' in parts[3:]:
                return f"gemm2halo_kernel<{epi}>"                                     # halo-staged conv3x3 (gemm4.hip)
            if parts[2] == "256x256" and parts[1] == "bf16" and not any(p.startswith("splitk") for p in parts[3:]):
                return f"gemm2pp_kernel<{t}, {bm}, {bn}, {wg}, {conv}, {epi}, 1>"     # ping-pong instantiation (gemm3.hip)
            return f"gemm2_kernel<{t}, {bm}, {bn}, {wg}, {conv}, {epi}>"
        return f"gemm_kernel<{t}, {bm}, {bn}, {conv}>"
    if parts[0] == "attention":
        t = "unsigned short" if parts[1] == "bf16" else "float"
        return f"attn_kernel<{t}, {parts[2][1:]}, false>"    # (third argument: the reference-tracking instantiation, tests only)
    return kind

def pmc_traffic(symbol):
    for name in PMC_TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            ent = json.load(open(path)).get(symbol)
            if ent is not None:
                return ent["hbm_bytes_per_launch"], name
    return None, None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json config (3: the single-GPU headline)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: what the config states; anything else is a rehearsal)")
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-roofline", action="store_true")
    ap.add_argument("--skip-fp32-latency", action="store_true")
    ap.add_argument("--skip-predrawn", action="store_true")
    ap.add_argument("--deterministic-plan", action="store_true", help="generate(deterministic_plan=True): bit-exact shard invariance, slower (rehearsal)")
    ap.add_argument("--gemm-impl", type=int, default=0, help="library switch word for same-box A/B runs (sdamd_set_gemm_impl; e.g. 8388608 = halo-staged convs off); marks the line a rehearsal")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as fresh child processes
    (torch.distributed.run, one per GPU) BEFORE this process touches the GPU, relay rank 0's JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    run(args)



def build_models(device, dtype, keep_cpu, with_encoder=False):
    torch.set_grad_enabled(False)
    specs = (("clip", clip.CLIP), ("decoder", decoder.VAE_Decoder), ("diffusion", diffusion.Diffusion))
    if with_encoder:
        from sd_amd import encoder
        specs += (("encoder", encoder.VAE_Encoder),)
    models, cpu_sd = {}, {}
    for name, cls in specs:
        m = synth.load_hashed(cls(), name + ".")
        if keep_cpu:
            cpu_sd[name] = {k: v.clone() for k, v in m.state_dict().items()}
        # text encoder on the f32 kernels, as pipeline.generate runs it (see pipeline.py)
        models[name] = engine.set_compute_dtype(m.to(device), torch.float32 if name == "clip" else dtype)
    return models, cpu_sd

class Txt2ImgJob:
    """Everything of generate() that can be prepared ahead, for a fixed batch of prompts/seeds."""

    def __init__(self, models, first_image, batch, denoise_steps, cfg_scale, device, use_graph=True, gather=True):
        self.models, self.B, self.gather = models, batch, gather
        ids = [synth.synthetic_token_ids(first_image + i) for i in range(batch)]
        self.pos = torch.tensor(ids, dtype=torch.long, device=device)
        self.neg = torch.tensor([synth.EMPTY_PROMPT_IDS] * batch, dtype=torch.long, device=device)
        seeds = [42 + first_image + i for i in range(batch)]
        ts, coef, noise, lat = pipeline.prepare_latents(seeds, pipeline.DDPMSampler, denoise_steps, (1, 4, 64, 64), device)
        self.latents0 = lat.float().contiguous()
        ctx = pipeline.encode_prompts(models["clip"], self.pos, self.neg)
        self.loop = pipeline.DenoiseLoop(models["diffusion"], ctx, ts, coef, noise, self.latents0, cfg_scale, True, use_graph)

    def step(self):
        ctx = pipeline.encode_prompts(self.models["clip"], self.pos, self.neg)
        self.loop.set_context(ctx)
        self.loop.reset(self.latents0)
        final = self.loop.run()
        images = pipeline.decode_to_uint8(self.models["decoder"], final)
        return distributed.all_gather_images(images) if self.gather else images   # (rank-local legs must not enter a collective)


def unet_step_ms(loop, latents0, iters=20, warm=3):
    """one captured Diffusion forward + CFG + step; median of >= 20 after 3 warm-ups, HIP events."""
    iters = max(1, min(iters, loop.n_steps - warm)) if loop.n_steps > warm else 1
    warm = min(warm, max(0, loop.n_steps - iters))
    loop.reset(latents0)
    for _ in range(warm):
        loop.step()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    evs[0].record()
    for i in range(iters):
        loop.step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    times = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(iters))
    loop.reset(latents0)
    return times[len(times) // 2]

def roofline_leg(call, device):
    """One eager (un-graphed) pass of the same generate() call with HIP-event pairs around every heavy launch.
    Rank 0 only: the call is made with gather=False (no collective in a rank-local leg)."""
    call(use_graph=False, gather=False)            # warm (weights already packed)
    torch.cuda.synchronize()
    ops.PROFILER = ops.LaunchTimer()
    try:
        call(use_graph=False, gather=False)
        summ = ops.PROFILER.summary()
    finally:
        ops.PROFILER = None
    is_mfma = lambda kind: kind.split("_")[0] in ("conv3x3", "gemm", "attention")
    rows = []
    for kind, v in summ.items():
        mfma = is_mfma(kind)
        dt = "bf16" if "_bf16" in kind else "f32"
        rate = v["work"] / (v["ms"] * 1e-3) / (1e12 if mfma else 1e9)
        rows.append({"kernel": kind, "launches": v["launches"], "total_ms": round(v["ms"], 3),
                     "avg_us": round(1e3 * v["ms"] / v["launches"], 2), "bound": "mfma" if mfma else "hbm",
                     "achieved": round(rate, 2), "unit": "TFLOP/s" if mfma else "GB/s",
                     "frac": round(rate / (PEAK[dt] if mfma else PEAK["hbm"]), 4)})
    rows.sort(key=lambda r: -r["total_ms"])
    # the roofline object is per kernel SYMBOL (what a rocprofv3 trace shows): split-K variants and all
    # problem shapes of one template instantiation are one entry
    by_sym = {}
    for kind, v in summ.items():
        e = by_sym.setdefault(kernel_symbol(kind), {"ms": 0.0, "work": 0.0, "bytes": 0.0, "launches": 0, "kinds": []})
        e["ms"] += v["ms"]; e["work"] += v["work"]; e["bytes"] += v["bytes"]; e["launches"] += v["launches"]; e["kinds"].append(kind)
    sym, top = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
    mfma = is_mfma(top["kinds"][0])
    dt = "bf16" if "_bf16" in top["kinds"][0] else "f32"
    peak = PEAK[dt] if mfma else PEAK["hbm"]
    rate = top["work"] / (top["ms"] * 1e-3) / (1e12 if mfma else 1e9)
    traffic, traffic_src = pmc_traffic(sym)
    roof = {"kernel": sym, "labels": sorted(top["kinds"]), "bound": "mfma" if mfma else "hbm",
            "achieved": round(rate, 2), "peak": peak, "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(rate / peak, 4),
            "traffic": traffic, "traffic_source": traffic_src, "launches": top["launches"],
            "avg_launch_us": round(1e3 * top["ms"] / top["launches"], 2),
            "algorithmic_work_per_launch": round(top["work"] / top["launches"] / (1e9 if mfma else 1e6), 3),
            "work_unit": "GFLOP" if mfma else "MB",
            "algorithmic_bytes_per_launch": round(top["bytes"] / top["launches"]),
            "note": "dominant kernel symbol of one bench step (most summed time over all its launches): mean algorithmic FLOPs "
                    "per launch / mean launch duration (HIP events on the launch stream, one extra eager pass; a split-K launch "
                    "includes its reduce kernel).  FLOPs are the reference op's 2*M*N*K (SURVEY.md §8d): the hi|lo operand "
                    "doubling and the 4/9-MAC upsample form earn no credit.  algorithmic_bytes = operands + weights + result "
                    "(+ residual), each once.  traffic = mean fabric bytes per launch from profiles/<traffic_source> (rocprofv3 "
                    "--pmc FETCH_SIZE / WRITE_SIZE passes over eager UNet steps, FETCH_SIZE doubled as MI355X_MICROARCH.md "
                    "prescribes for gfx950), null if not collected for this symbol"}
    return roof, rows

def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"

def cpu_baseline_leg(cpu_sd):
    """BASELINE config 1 (txt2img b1, 50-step DDPM, CFG 7.5, fp32, CPU) on the CPU oracle, bounded: the first 2 of
    the 50 UNet steps (UNet batch 2, sampler step included), 1 VAE decode, the 2 CLIP encodes -- each stage timed,
    then summed to the full run, once as this framework runs it (final decode only) and once as the reference runs
    it (preview decode after every step, pipeline.py:156-164)."""
    from oracle import sd_oracle as O
    threads = torch.get_num_threads()
    toks = torch.tensor([synth.synthetic_token_ids(0), synth.EMPTY_PROMPT_IDS])
    t0 = time.perf_counter()
    ctx = O.clip_forward(cpu_sd["clip"], toks)
    t_clip2 = time.perf_counter() - t0
    samp = O.DDPMSampler(torch.Generator().manual_seed(42))
    samp.set_steps(50)
    lat = torch.randn((1, 4, 64, 64), generator=samp.generator)
    t_unet = []
    for t in samp.timesteps[:2]:
        t0 = time.perf_counter()
        out = O.diffusion_forward(cpu_sd["diffusion"], lat.repeat(2, 1, 1, 1), ctx, O.get_time_embedding(int(t)))
        eps = 7.5 * (out[0:1] - out[1:2]) + out[1:2]
        lat = samp.step(int(t), lat, eps)
        t_unet.append(time.perf_counter() - t0)
    t_unet = sum(t_unet) / len(t_unet)
    t0 = time.perf_counter()
    O.vae_decode(cpu_sd["decoder"], lat)
    t_dec = time.perf_counter() - t0
    per_image = 50 * t_unet + t_dec + t_clip2
    per_image_preview = 50 * (t_unet + t_dec) + t_clip2
    return {"value": round(1.0 / per_image, 6), "unit": "images/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
            "value_with_preview_decode": round(1.0 / per_image_preview, 6),
            "sample": f"BASELINE config 1 bounded: steps 1-2 of 50 (UNet b2@64x64 fp32 + CFG + DDPM step, mean {t_unet:.2f}s) + 1 VAE "
                      f"decode ({t_dec:.2f}s) + 2 CLIP encodes ({t_clip2:.2f}s) on {threads} threads; value = 1 / (50 steps + final "
                      f"decode + CLIP) = {per_image:.0f}s per image; value_with_preview_decode = the reference's loop with its "
                      f"per-step preview decode = {per_image_preview:.0f}s per image",
            "unet_step_s": round(t_unet, 3), "vae_decode_s": round(t_dec, 3), "clip2_s": round(t_clip2, 3)}

class GenerateCall:
    """This rank's share of a BASELINE config as ONE pipeline.generate() call (prompts, seeds, input images fixed)."""

    def __init__(self, models, cfg, first_image, batch, denoise_steps, device, deterministic_plan=False):
        self.models, self.cfg, self.B, self.device = models, cfg, batch, device
        self.det = deterministic_plan
        self.tok = synth.SyntheticTokenizer()
        self.prompts = [f"synthetic:{first_image + i}" for i in range(batch)]
        self.kw = dict(models=models, seed=42 + first_image, device=str(device), sampler="ddpm", steps=denoise_steps,
                       cfg_scale=7.5, tokenizer=self.tok, compute_dtype=torch.bfloat16)
        if cfg["img2img"]:   # input image = uint8 uniform noise, seed 7 + i; strength 0.8 as intended
            import numpy as np
            from PIL import Image
            r = cfg["res"]
            self.kw.update(input_image=[Image.fromarray(np.random.default_rng(7 + first_image + i).integers(0, 256, (r, r, 3), dtype=np.uint8))
                                        for i in range(batch)], strength=0.8, strength_semantics="intended")

    def __call__(self, use_graph=True, gather=True):
        return pipeline.generate(self.prompts, "", use_graph=use_graph, all_gather=gather, deterministic_plan=self.det, **self.kw)

    def loop(self):
        """The captured DenoiseLoop generate() keeps for this shape (for the UNet step latency)."""
        return next(reversed(pipeline._LOOP_CACHE.values()))[1]


# ---------------------------------------------------------------------------------------------------------------
def run(args):
    rank, world, local = distributed.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dtype = torch.bfloat16
    cfg = CONFIGS[args.config]
    if cfg["global_batch"] is not None:            # strong scaling: the job's images sharded over the ranks
        total = cfg["global_batch"] if args.batch is None else args.batch * world
        lo, hi = distributed.shard_range(total, rank, world)
        first, batch = lo, hi - lo
        if total % world:
            raise SystemExit(f"config {args.config}: global batch {total} does not divide over {world} GPUs")
    else:                                          # weak scaling: every GPU its own images
        batch = cfg["batch"] if args.batch is None else args.batch
        first, total = rank * batch, world * batch
    standard = args.batch is None and args.denoise_steps == 50 and not args.deterministic_plan and not args.gemm_impl
    if args.gemm_impl:
        from sd_amd import _lib
        _lib.lib().sdamd_set_gemm_impl(args.gemm_impl)
    if cfg["res"] != 512:                          # the reference's resolution is four module constants (pipeline.py:9-13)
        pipeline.WIDTH = pipeline.HEIGHT = cfg["res"]
        pipeline.LATENTS_WIDTH = pipeline.LATENT_HEIGHT = cfg["res"] // 8
    models, cpu_sd = build_models(device, dtype, keep_cpu=(rank == 0 and world == 1 and not args.skip_cpu_baseline),
                                  with_encoder=cfg["img2img"])

    if os.environ.get("SDAMD_DUMP_MAPS"):   # profiling runs: which module every PC of this process belongs to (all libraries are loaded by now)
        try:
            with open("/proc/self/maps") as src, open(os.path.join(os.environ["SDAMD_DUMP_MAPS"], f"maps_{os.getpid()}.txt"), "w") as dst:
                dst.write(src.read())
        except OSError:
            pass
    call = GenerateCall(models, cfg, first, batch, args.denoise_steps, device, args.deterministic_plan)
    for _ in range(max(1, args.warmup)):           # (the first call captures the graphs: at least one untimed call)
        call()
    torch.cuda.synchronize()
    distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = call()
    torch.cuda.synchronize()
    distributed.barrier()
    elapsed = distributed.max_over_ranks(time.perf_counter() - t0, device)
    assert out.shape == (total, cfg["res"], cfg["res"], 3) and out.dtype.name == "uint8", out.shape

    if rank != 0:
        return
    loop = call.loop()
    ms_per_step = 1e3 * elapsed / args.steps
    value = total * args.steps / elapsed
    n_denoise = loop.n_steps
    result = {
        "metric": f"images/sec @{cfg['res']}x{cfg['res']} 50-step DDPM+CFG", "value": round(value, 4), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": round(ms_per_step, 2),
        "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": (f"BASELINE config {args.config}: " if standard else f"NON-STANDARD rehearsal of config {args.config}: ") +
                               cfg["what"] + f"; {batch} images on this GPU, {n_denoise} UNet steps per image; one step = wall(pipeline.generate): "
                               "tokenise + CLIP encode + host noise draw and upload + the step loop as one hipGraph launch + final VAE "
                               "decode + uint8 (+ the all-gather when N>1) + device->host copy; hash-init SD-v1.5 weights, synthetic token ids",
                   "global_batch": total, "per_gpu_batch": batch, "denoise_steps": n_denoise, "resolution": cfg["res"],
                   "parallelism": f"image-shard x{world}"},
        "tflops_effective": round(value * cfg["gflop"] / 1e3, 1),
        "frac_of_bf16_mfma_peak": round(value * cfg["gflop"] / 1e3 / (PEAK["bf16"] * world), 4),
    }
    extra = {"unet_step_ms_bf16": round(unet_step_ms(loop, loop.latents.clone()), 3), "unet_batch_bf16": 2 * batch,
             "unet_step_latency_method": "median of 20 one-step graph replays after 3 warm-ups, HIP events per replay",
             "whole_loop_graph": bool(pipeline.WHOLE_LOOP_GRAPH and loop.full_graph is not None)}
    if not args.skip_roofline:
        roof, rows = roofline_leg(call, device)
        result["roofline"] = roof
        extra["kernel_breakdown"] = rows[:12]
        # north_star: achieved HBM GB/s of the fused norm and sampler kernels against the 8 TB/s peak
        extra["hbm_kernels"] = [r for r in rows if r["bound"] == "hbm"]
    pipeline.clear_loop_cache()
    del loop
    if world == 1 and args.config == 3 and not args.skip_predrawn:
        # the same workload with everything host-side hoisted (noise pre-drawn and resident, token ids resident): what the
        # headline would be if generate()'s host work were free
        job = Txt2ImgJob(models, 0, batch, args.denoise_steps, 7.5, device)
        job.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            job.step()
        torch.cuda.synchronize()
        extra["predrawn_noise_images_per_s"] = round(2 * batch / (time.perf_counter() - t0), 4)
        extra["predrawn_noise_note"] = ("CLIP + step loop (one graph launch) + VAE decode + uint8 with noise, latents and token ids already "
                                        "in HBM and no device->host copy: the headline minus generate()'s host-side work")
        del job
    if world == 1 and args.config == 3 and not args.skip_fp32_latency:   # BASELINE config 2: UNET step, UNet batch 2, fp32 (and the same in bf16 mode)
        j1 = Txt2ImgJob(models, 0, 1, args.denoise_steps, 7.5, device)
        extra["unet_step_ms_bf16_b2"] = round(unet_step_ms(j1.loop, j1.latents0), 3)
        del j1
        engine.set_compute_dtype(models["diffusion"], torch.float32)
        # fp32 operands, both ways of reaching the matrix cores (include/sdamd.h sdamd_set_f32_mode): the default forms
        # each f32 product from three-term bf16 splits (six bf16 MFMAs, f32 accumulate; as close to float64 as the f32
        # MFMA, tests/test_hip_ops.py::test_f32_modes_against_float64), "exact" is v_mfma_f32_16x16x4_f32
        for mode, key in (("bf16x3", "unet_step_ms_fp32_b2"), ("exact", "unet_step_ms_fp32_b2_exact_mfma")):
            ops.set_f32_mode(mode)
            j2 = Txt2ImgJob(models, 0, 1, args.denoise_steps, 7.5, device)   # (a captured loop keeps the mode it was captured under)
            extra[key] = round(unet_step_ms(j2.loop, j2.latents0), 3)
            del j2
        ops.set_f32_mode("bf16x3")
        extra["unet_step_fp32_b2_tflops"] = round(1.6065 / (extra["unet_step_ms_fp32_b2"] * 1e-3), 1)
        extra["unet_step_fp32_b2_note"] = ("BASELINE config 2 (UNET step, latent 64x64x4, batch 2 = CFG, fp32 tensors): f32 in / f32 out / f32 "
                                           "accumulate; operands split exactly into 3 bf16 terms in registers, 6 MFMA products per f32 product; "
                                           "_exact_mfma = the same step on the f32 MFMA instruction (157 TFLOP/s peak)")
        engine.set_compute_dtype(models["diffusion"], dtype)
    if cpu_sd:
        result["cpu_baseline"] = cpu_baseline_leg(cpu_sd)
    result["extra"] = extra
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
