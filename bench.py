#!/usr/bin/env python3
"""Headline benchmark: images/sec @512x512, 50-step DDPM + CFG (BASELINE.json), one process per GPU.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch per GPU: CLIP encode of 2B prompts ->
50 x [UNet forward on 2B latents + fused CFG/DDPM step] (replayed hipGraph) -> VAE decode -> uint8
-> (N>1) one RCCL all-gather of the images.  Workload = BASELINE config 3 (txt2img, bf16, batch 8
per GPU); weak scaling: every GPU does its own 8 images.  Token ids, noise and weights are resident
in HBM before the timed region; weight hashing / repacking / graph capture happen in warm-up.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the dominant kernel family of the timed workload, timed live with HIP events on
                the launch stream in one extra eager (un-graphed) pass of the same step,
  cpu_baseline  the CPU oracle (oracle/sd_oracle.py, a port) timed on this host's cores on a
                bounded sample (1 UNet step + 1 VAE decode + 2 CLIP encodes) and extrapolated,
  extra         UNet step latencies (bf16 at the bench batch, fp32 at UNet batch 2 = config 2)
                and the per-kernel-family breakdown.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sd_amd  # noqa: E402,F401
from sd_amd import clip, decoder, diffusion, distributed, engine, ops, pipeline, synth  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3, "hbm": 8000.0}  # MI355X_MICROARCH.md: dense TFLOP/s, GB/s
GFLOP_PER_IMAGE = 82866.0                              # 50 x 1606.5 + 2514.5 + 26.6


def build_models(device, dtype, keep_cpu):
    torch.set_grad_enabled(False)
    specs = (("clip", clip.CLIP), ("decoder", decoder.VAE_Decoder), ("diffusion", diffusion.Diffusion))
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

    def __init__(self, models, first_image, batch, denoise_steps, cfg_scale, device, use_graph=True):
        self.models, self.B = models, batch
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
        return distributed.all_gather_images(images)


def unet_step_ms(loop, latents0, iters=10):
    loop.reset(latents0)
    loop.step()
    torch.cuda.synchronize()
    loop.reset(latents0)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        loop.step()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def roofline_leg(models, args, device):
    """One eager pass of the same step with HIP-event pairs around every heavy launch."""
    job = Txt2ImgJob(models, 0, args.batch, args.denoise_steps, 7.5, device, use_graph=False)
    job.step()                                     # warm (weights already packed)
    torch.cuda.synchronize()
    ops.PROFILER = ops.LaunchTimer()
    try:
        job.step()
        summ = ops.PROFILER.summary()
    finally:
        ops.PROFILER = None
    rows = []
    for kind, v in summ.items():
        mfma = kind.split("_")[0] in ("conv3x3", "gemm", "attention")
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
        e = by_sym.setdefault(kernel_symbol(kind), {"ms": 0.0, "work": 0.0, "launches": 0, "kinds": []})
        e["ms"] += v["ms"]; e["work"] += v["work"]; e["launches"] += v["launches"]; e["kinds"].append(kind)
    sym, top = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
    mfma = top["kinds"][0].split("_")[0] in ("conv3x3", "gemm", "attention")
    dt = "bf16" if "_bf16" in top["kinds"][0] else "f32"
    peak = PEAK[dt] if mfma else PEAK["hbm"]
    rate = top["work"] / (top["ms"] * 1e-3) / (1e12 if mfma else 1e9)
    roof = {"kernel": sym, "labels": sorted(top["kinds"]), "bound": "mfma" if mfma else "hbm",
            "achieved": round(rate, 2), "peak": peak, "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(rate / peak, 4),
            "traffic": pmc_traffic(sym), "launches": top["launches"],
            "avg_launch_us": round(1e3 * top["ms"] / top["launches"], 2),
            "algorithmic_work_per_launch": round(top["work"] / top["launches"] / (1e9 if mfma else 1e6), 3),
            "work_unit": "GFLOP" if mfma else "MB",
            "note": "dominant kernel symbol of one bench step (most summed time over all its launches): mean algorithmic FLOPs per "
                    "launch / mean launch duration, HIP events on the launch stream in one extra eager pass; traffic = mean HBM bytes "
                    "per launch from profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over eager UNet "
                    "steps, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), null if not collected"}
    return roof, rows


def kernel_symbol(kind):
    """Bench label -> substring of the kernel's demangled name in a rocprofv3 trace."""
    parts = kind.split("_")
    if parts[0] in ("gemm", "conv3x3") and len(parts) >= 3 and "x" in parts[2]:
        bm, bn = parts[2].split("x")
        conv = "true" if parts[0] == "conv3x3" else "false"
        t = "unsigned short" if parts[1] == "bf16" else "float"
        wg = {"128x320": "2, 4, 2", "128x160": "2, 2, 2", "256x256": "2, 4, 2"}.get(parts[2])
        if wg is not None:      # 8-wave / 4-wave LDS-DMA family (gemm2.hip); last argument: 1 = residual-prefetch, 2 = GeGLU-only epilogue
            return f"gemm2_kernel<{t}, {bm}, {bn}, {wg}, {conv}, {1 if 'rp' in parts[3:] else 2 if 'geglu' in parts[3:] else 0}>"
        return f"gemm_kernel<{t}, {bm}, {bn}, {conv}>"
    if parts[0] == "attention":
        t = "unsigned short" if parts[1] == "bf16" else "float"
        return f"attn_kernel<{t}, {parts[2][1:]}>"
    return kind


def pmc_traffic(symbol):
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    ent = json.load(open(path)).get(symbol)
    return None if ent is None else ent["hbm_bytes_per_launch"]


def cpu_baseline_leg(cpu_sd):
    """CPU oracle (a port of the reference path) on a bounded sample, extrapolated to 1 image."""
    from oracle import sd_oracle as O
    threads = torch.get_num_threads()
    lat = torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(0))
    toks = torch.tensor([synth.synthetic_token_ids(0), synth.EMPTY_PROMPT_IDS])
    t0 = time.perf_counter()
    ctx = O.clip_forward(cpu_sd["clip"], toks)
    t_clip2 = time.perf_counter() - t0
    temb = O.get_time_embedding(980)
    t0 = time.perf_counter()
    O.diffusion_forward(cpu_sd["diffusion"], lat, ctx, temb)
    t_unet = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.vae_decode(cpu_sd["decoder"], lat[:1])
    t_dec = time.perf_counter() - t0
    per_image = 50 * t_unet + t_dec + t_clip2
    return {"value": round(1.0 / per_image, 6), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 UNet step b2@64x64 fp32 ({t_unet:.2f}s) + 1 VAE decode ({t_dec:.2f}s) + 2 CLIP encodes "
                      f"({t_clip2:.2f}s) on {threads} threads, extrapolated to 50 steps + final decode only",
            "unet_step_s": round(t_unet, 3), "vae_decode_s": round(t_dec, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (config 3: 8)")
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-roofline", action="store_true")
    ap.add_argument("--skip-fp32-latency", action="store_true")
    args = ap.parse_args()

    rank, world, local = distributed.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dtype = torch.bfloat16
    models, cpu_sd = build_models(device, dtype, keep_cpu=(rank == 0 and world == 1 and not args.skip_cpu_baseline))

    job = Txt2ImgJob(models, rank * args.batch, args.batch, args.denoise_steps, 7.5, device)
    for _ in range(args.warmup):
        job.step()
    torch.cuda.synchronize()
    distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = job.step()
    torch.cuda.synchronize()
    distributed.barrier()
    elapsed = distributed.max_over_ranks(time.perf_counter() - t0, device)
    assert out.shape[0] == world * args.batch and out.dtype == torch.uint8

    if rank != 0:
        return
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.batch * args.steps / elapsed
    result = {
        "metric": "images/sec @512x512 50-step DDPM+CFG", "value": round(value, 4), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE config 3: txt2img 512x512, 50-step DDPM, CFG 7.5, bf16, batch 8 per GPU, "
                               "incl. CLIP encode + final VAE decode + uint8 (+ all-gather when N>1); hash-init SD-v1.5 weights, "
                               "synthetic token ids",
                   "global_batch": world * args.batch, "per_gpu_batch": args.batch, "denoise_steps": args.denoise_steps,
                   "resolution": 512, "parallelism": f"image-shard x{world}"},
        "tflops_effective": round(value * GFLOP_PER_IMAGE / 1e3, 1),
        "frac_of_bf16_mfma_peak": round(value * GFLOP_PER_IMAGE / 1e3 / (PEAK["bf16"] * world), 4),
    }
    extra = {"unet_step_ms_bf16": round(unet_step_ms(job.loop, job.latents0), 3), "unet_batch_bf16": 2 * args.batch}
    if not args.skip_roofline:
        roof, rows = roofline_leg(models, args, device)
        result["roofline"] = roof
        extra["kernel_breakdown"] = rows[:12]
        # north_star: achieved HBM GB/s of the fused norm and sampler kernels against the 8 TB/s peak
        extra["hbm_kernels"] = [r for r in rows if r["bound"] == "hbm"]
    if world == 1 and not args.skip_fp32_latency:   # BASELINE config 2: UNET step, UNet batch 2, fp32 (and the same in bf16 mode)
        j1 = Txt2ImgJob(models, 0, 1, args.denoise_steps, 7.5, device)
        extra["unet_step_ms_bf16_b2"] = round(unet_step_ms(j1.loop, j1.latents0), 3)
        del j1
        engine.set_compute_dtype(models["diffusion"], torch.float32)
        j2 = Txt2ImgJob(models, 0, 1, args.denoise_steps, 7.5, device)
        extra["unet_step_ms_fp32_b2"] = round(unet_step_ms(j2.loop, j2.latents0), 3)
        extra["unet_step_fp32_b2_tflops"] = round(1.6065 / (extra["unet_step_ms_fp32_b2"] * 1e-3), 1)
        del j2
        engine.set_compute_dtype(models["diffusion"], dtype)
    if cpu_sd:
        result["cpu_baseline"] = cpu_baseline_leg(cpu_sd)
    result["extra"] = extra
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
