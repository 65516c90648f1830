"""CPU-only checks: the C-ABI library loads and exports every symbol include/sdamd.h declares,
host-side schedules of the product equal the reference's goldens, synthetic inputs are stable,
and the product refuses to run without the HIP device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import sd_amd
from sd_amd import _lib, ops, pipeline, synth
from sd_amd.sampler.ddpm import DDPMSampler
from conftest import golden, ROOT


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sdamd.h")).read()
    declared = set(re.findall(r"\b(sdamd_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().sdamd_abi_version() == 7
    # host-side tile choice (no GPU needed): 8-wave LDS-DMA tile for a big f32 GEMM, 4-wave 64x64 for a small one
    assert _lib.lib().sdamd_gemm_tile(65536, 320, 320, 1, 0, 0, 0) == 128320
    assert _lib.lib().sdamd_gemm_tile(512, 1280, 1280, 1, 0, 0, 0) == 64064


def test_product_schedules_match_reference_bit_exactly():
    g = golden("sampler_tables")
    s = DDPMSampler(torch.Generator().manual_seed(0))
    assert np.array_equal(s.alpha_cumprod.numpy(), g["alpha_cumprod"]) and np.array_equal(s.betas.numpy(), g["betas"])
    s.set_steps(50)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps50"]) and s.timesteps.dtype == torch.int64
    temb = torch.cat([pipeline.get_time_embedding(t) for t in s.timesteps])
    assert np.array_equal(temb.numpy(), g["time_emb50"])
    s.set_strength(0.99)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps50_strength099"])
    s2 = DDPMSampler(torch.Generator().manual_seed(0)); s2.set_steps(50); s2.set_strength(0.8)
    assert len(s2.timesteps) == 0                      # literal reference semantics (its bug)
    s3 = DDPMSampler(torch.Generator().manual_seed(0)); s3.set_steps(50); s3.set_strength_steps(0.8)
    assert len(s3.timesteps) == 40 and int(s3.timesteps[0]) == 780
    s4 = DDPMSampler(torch.Generator().manual_seed(0)); s4.set_steps(20)
    assert np.array_equal(s4.timesteps.numpy(), g["timesteps20"])


def test_coefficient_table_matches_oracle_and_shapes():
    from oracle import sd_oracle as O
    s = DDPMSampler(torch.Generator().manual_seed(0)); s.set_steps(50)
    o = O.DDPMSampler(torch.Generator().manual_seed(0)); o.set_steps(50)
    tab = s.coefficient_table()
    assert tab.shape == (50, 8) and tab.dtype == torch.float32
    for i, t in enumerate(o.timesteps):
        c = o.coefficients(t)
        assert all(float(tab[i][j]) == float(c[j]) for j in range(4))
        assert float(tab[i][5]) == (1.0 if int(t) > 0 else 0.0)
        if int(t) > 0:
            assert float(tab[i][4]) == float(c[4])


def test_rng_call_order_of_prepare_latents():
    """txt2img: randn(latents) then one draw per step with t>0, per image, from a CPU generator seeded seed+i."""
    ts, coef, noise, lat = pipeline.prepare_latents([42, 43], DDPMSampler, 3, (1, 4, 8, 8), torch.device("cpu"))
    for b, seed in enumerate((42, 43)):
        g = torch.Generator().manual_seed(seed)
        assert torch.equal(lat[b], torch.randn((4, 8, 8), generator=g))
        for si, t in enumerate(ts):
            if int(t) > 0:
                assert torch.equal(noise[si, b], torch.randn((4, 8, 8), generator=g))
            else:
                assert float(noise[si, b].abs().sum()) == 0.0
    assert coef.shape == (3, 8)


@pytest.mark.parametrize("shape", [(1, 4, 64, 64), (1, 4, 3, 3), (1, 4, 2, 2)])
def test_batched_step_noise_equals_draw_by_draw(shape):
    """prepare_latents takes an image's per-step draws as one randn (threads over images): same stream as the
    reference's call-by-call draws (ddpm.py:82), also on the sizes that take the sequential fallback (numel % 16 != 0)."""
    seeds = [7, 8, 9]
    ts, _, noise, lat = pipeline.prepare_latents(seeds, DDPMSampler, 5, shape, torch.device("cpu"))
    for b, seed in enumerate(seeds):
        g = torch.Generator().manual_seed(seed)
        assert torch.equal(lat[b], torch.randn(shape[1:], generator=g))
        for si, t in enumerate(ts):
            want = torch.randn(shape[1:], generator=g) if int(t) > 0 else torch.zeros(shape[1:])
            assert torch.equal(noise[si, b], want), (b, si)


def test_synthetic_inputs_are_stable():
    ids = synth.synthetic_token_ids(0)
    assert len(ids) == 77 and ids[0] == synth.BOS and ids[-1] == synth.EOS and ids[:4] == [49406, 45205, 21483, 39265]
    assert np.array_equal(golden("clip")["tokens"][0], np.array(ids))
    tok = synth.SyntheticTokenizer().batch_encode_plus(["synthetic:0", ""], padding="max_length", max_length=77).input_ids
    assert tok[0] == ids and tok[1] == synth.EMPTY_PROMPT_IDS
    t = synth.hash_tensor("a.weight", (4, 8))
    assert abs(float(t[0, 0]) - (-0.4404)) < 1e-4 and t.dtype == torch.float32


def test_no_cpu_fallback():
    with pytest.raises(_lib.SdamdError, match="HIP device"):
        ops.layernorm(torch.zeros(4, 320), torch.ones(320), torch.zeros(320))
    from sd_amd import diffusion
    blk = diffusion.UNET_ResidualBlock(64, 64)
    with pytest.raises(_lib.SdamdError):
        blk(torch.zeros(1, 64, 4, 4), torch.zeros(1, 1280))
    with pytest.raises(_lib.SdamdError, match="HIP device"):
        pipeline.generate("synthetic:0", "", models={}, device="cpu", tokenizer=synth.SyntheticTokenizer())
    with pytest.raises(ValueError, match="Strength must be between 0 and 1"):
        pipeline.generate("a", "", strength=1.5, models={}, device="cuda", tokenizer=synth.SyntheticTokenizer())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "stable-diffusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                for line in open(os.path.join(dirpath, f)):
                    if re.match(r"\s*(from|import)\s", line):
                        assert "oracle" not in line, (f, line)


def test_bench_labels_map_to_kernel_symbols():
    """bench.py attributes profiler labels to rocprofv3 kernel names (roofline object, PMC traffic lookup): the last
    template argument of gemm2_kernel is its epilogue instantiation (1 = residual prefetch, label suffix _rp)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    ks = bench.kernel_symbol
    assert ks("conv3x3_bf16_128x160") == "gemm2_kernel<unsigned short, 128, 160, 2, 2, 2, true, 0>"
    assert ks("conv3x3_bf16_128x160_splitk2") == "gemm2_kernel<unsigned short, 128, 160, 2, 2, 2, true, 0>"
    assert ks("conv3x3_bf16_128x160_rp") == "gemm2_kernel<unsigned short, 128, 160, 2, 2, 2, true, 1>"
    assert ks("gemm_bf16_128x320_rp") == "gemm2_kernel<unsigned short, 128, 320, 2, 4, 2, false, 1>"
    assert ks("gemm_bf16_256x256_geglu") == "gemm2pp_kernel<unsigned short, 256, 256, 2, 4, 2, false, 2, 1>"
    assert ks("conv3x3_bf16_256x256") == "gemm2pp_kernel<unsigned short, 256, 256, 2, 4, 2, true, 0, 1>"
    assert ks("conv3x3_bf16_128x160_halo") == "gemm2halo_kernel<0>" and ks("conv3x3_bf16_128x160_rp_halo") == "gemm2halo_kernel<1>"
    assert ks("conv3x3_bf16_128x160_splitk2_halo") == "gemm2halo_kernel<0>"
    assert ks("conv3x3_bf16_128x160_hbw") == ks("conv3x3_bf16_128x160_rp_hbw") == "gemm2hbw_kernel"
    assert ks("conv3x3_bf16_128x160_hbwup") == "gemm2hbwup_kernel"
    assert ks("gemm_bf16_128x320_ln") == "gemm2_kernel<unsigned short, 128, 320, 2, 4, 2, false, 3>"
    assert ks("gemm_f32_128x160") == "gemm2_kernel<float, 128, 160, 2, 2, 2, false, 0>"
    assert ks("gemm_f32_64x64") == "gemm_kernel<float, 64, 64, false>"
    assert ks("conv3x3_bf16_256x320") == "gemm2_kernel<unsigned short, 256, 320, 4, 2, 2, true, 0>"
    assert ks("attention_bf16_d40") == "attn_kernel<unsigned short, 40, false>"
    assert ks("groupnorm_f32") == "groupnorm_f32"
    traffic, source = bench.pmc_traffic("gemm2_kernel<unsigned short, 128, 160, 2, 2, 2, true, 0>")
    assert (traffic is None and source is None) or (traffic > 1e8 and source.endswith("_pmc_traffic.json"))        # profiles/r01_pmc_traffic.json carries the dominant kernel under this name


def test_bench_rank_local_legs_do_not_enter_collectives():
    """bench.py with N > 1: after the timed region every rank but 0 returns; the legs rank 0 then runs alone (roofline,
    latency) must not call the all-gather (a 2-rank rehearsal crashed exactly there before `gather=False` existed)."""
    import inspect
    import bench
    assert "gather=False" in inspect.getsource(bench.roofline_leg)
    src = inspect.getsource(bench.Txt2ImgJob.step)
    assert "if self.gather else images" in src

