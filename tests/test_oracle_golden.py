"""Pins the CPU oracle (oracle/sd_oracle.py) to outputs of the reference itself
(tests/golden/*.npz, produced by tools/gen_golden.py from /root/reference/src)."""
import numpy as np
import pytest
import torch

import sd_amd
from sd_amd import synth
from oracle import sd_oracle as O
from conftest import golden, rel_l2, hin

torch.set_grad_enabled(False)
TOL = 1e-6


def hashed(shapes, prefix):
    return synth.hash_state_dict(shapes, prefix)


def test_sampler_tables_bit_exact():
    g = golden("sampler_tables")
    s = O.DDPMSampler(torch.Generator().manual_seed(0))
    assert np.array_equal(s.alpha_cumprod.numpy(), g["alpha_cumprod"])
    assert np.array_equal(s.betas.numpy(), g["betas"])
    s.set_steps(50)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps50"])
    temb = torch.cat([O.get_time_embedding(t) for t in s.timesteps])
    assert np.array_equal(temb.numpy(), g["time_emb50"])
    s.set_strength(0.99)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps50_strength099"])
    s20 = O.DDPMSampler(torch.Generator().manual_seed(0)); s20.set_steps(20)
    assert np.array_equal(s20.timesteps.numpy(), g["timesteps20"])
    s08 = O.DDPMSampler(torch.Generator().manual_seed(0)); s08.set_steps(50); s08.set_strength(0.8)
    assert len(s08.timesteps) == int(g["timesteps50_strength08_len"]) == 0   # reference bug, SURVEY §3.4


def test_ddpm_step_bit_exact():
    g = golden("ddpm_step")
    for steps, ts in ((50, (980, 500, 20, 0)), (20, (950, 0))):
        for t in ts:
            sm = O.DDPMSampler(torch.Generator().manual_seed(1234 + t)); sm.set_steps(steps)
            lat, eps = hin(f"ddpm.lat.{t}", (1, 4, 16, 16), 2.0), hin(f"ddpm.eps.{t}", (4, 16, 16), 1.5)
            noise = torch.from_numpy(g[f"noise{steps}_{t}"]) if t > 0 else None
            # bit-exact given the reference's own noise draw ...
            assert np.array_equal(sm.step(torch.tensor(t), lat, eps, noise=noise).numpy(), g[f"step{steps}_{t}"]), (steps, t)
            # ... and the generator is consumed in the same order/shape (ulp-level host ISA differences allowed)
            assert rel_l2(sm.step(torch.tensor(t), lat, eps), g[f"step{steps}_{t}"]) < 1e-6
    sm = O.DDPMSampler(torch.Generator().manual_seed(77)); sm.set_steps(50)
    x = hin("ddpm.an", (1, 4, 16, 16))
    assert np.array_equal(sm.add_noise(x, torch.tensor(780), noise=torch.from_numpy(g["add_noise_780_noise"])).numpy(), g["add_noise_780"])
    assert rel_l2(sm.add_noise(x, torch.tensor(780)), g["add_noise_780"]) < 1e-6


def _attn_shapes(dim, bias_in, cross=False):
    if cross:
        s = {"q_proj.weight": (dim, dim), "k_proj.weight": (dim, 768), "v_proj.weight": (dim, 768)}
    else:
        s = {"in_proj.weight": (3 * dim, dim)}
        if bias_in:
            s["in_proj.bias"] = (3 * dim,)
    s["out_proj.weight"] = (dim, dim)
    s["out_proj.bias"] = (dim,)
    return s


def test_attention_ops():
    g = golden("attention_ops")
    for tag, heads, dim, seq, causal in (("h8d40", 8, 320, 64, False), ("h8d80", 8, 640, 48, False),
                                         ("h8d160", 8, 1280, 16, False), ("clip", 12, 768, 77, True),
                                         ("vae", 1, 512, 64, False)):
        sd = hashed(_attn_shapes(dim, tag in ("clip", "vae")), f"op.self.{tag}.")
        sd = {"a." + k: v for k, v in sd.items()}
        out = O.self_attention(sd, "a", hin(f"self.{tag}", (2, seq, dim)), heads, causal)
        assert rel_l2(out, g["self_" + tag]) < TOL, tag
    for tag, heads, dim, seq in (("h8d40", 8, 320, 64), ("h8d160", 8, 1280, 16)):
        sd = hashed(_attn_shapes(dim, False, cross=True), f"op.cross.{tag}.")
        sd = {"a." + k: v for k, v in sd.items()}
        out = O.cross_attention(sd, "a", hin(f"cross.x.{tag}", (2, seq, dim)), hin(f"cross.y.{tag}", (2, 77, 768)), heads)
        assert rel_l2(out, g["cross_" + tag]) < TOL, tag
