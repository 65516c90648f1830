"""Checkpoint converter / loader against the reference's own converter.

tests/golden/converter_map.json was produced by tools/gen_golden.py running the reference's
``model_converter.load_from_standard_weights`` on a synthetic CompVis-layout checkpoint (hash values,
tiny shapes): it lists the checkpoint keys + shapes that went in and shape + crc32 of every tensor
that came out.  The tests regenerate the same checkpoint and require bit-identical output.
"""
import json
import os
import zlib

import numpy as np
import pytest
import torch

import sd_amd  # noqa: F401
from sd_amd import model_converter as mc, synth
from conftest import GOLDEN, golden


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN, "converter_map.json")) as f:
        return json.load(f)


def synthetic_checkpoint(gold):
    return {k: torch.from_numpy(synth.hash_uniform("ckpt:" + k, int(np.prod(s))).reshape(s))
            for k, s in gold["source"].items()}


def check_against_golden(conv, gold):
    assert set(conv) == set(gold["converted"]) == set(mc.MODELS)
    for model, want in gold["converted"].items():
        assert set(conv[model]) == set(want), model
        for k, w in want.items():
            t = conv[model][k]
            assert list(t.shape) == w["shape"], (model, k)
            assert zlib.crc32(t.contiguous().numpy().tobytes()) == w["crc32"], (model, k)


def test_rules_read_exactly_the_keys_the_reference_reads(gold):
    assert set(mc.source_keys()) == set(gold["source"])
    assert len(mc.source_keys()) == 1130
    # q|k|v concatenations: 16 UNet self-attention weights, 2 VAE (weight + bias), 12 CLIP (weight + bias)
    assert sum(1 for *_, srcs, _ in mc.conversion_rules() if len(srcs) == 3) == 16 + 2 * 2 + 12 * 2


def test_rules_produce_the_reference_state_dict_abi():
    """Destination keys == the reference modules' state_dict keys (loader.py:18-27 loads strictly)."""
    g = golden("state_dict_keys")
    by_model = {m: set() for m in mc.MODELS}
    for model, dst, _, _ in mc.conversion_rules():
        assert dst not in by_model[model], dst
        by_model[model].add(dst)
    for m in mc.MODELS:
        assert by_model[m] == set(g[m + "_keys"].tolist()), m
    assert sum(len(v) for v in by_model.values()) == 1042


@pytest.mark.parametrize("container", ["safetensors", "ckpt_state_dict", "ckpt_bare", "memory"])
def test_conversion_is_bit_identical_to_the_reference(gold, tmp_path, container):
    ckpt = synthetic_checkpoint(gold)
    ckpt["model_ema.decay"] = torch.zeros(1)          # entries the conversion never reads
    if container == "safetensors":
        from safetensors.torch import save_file
        path = str(tmp_path / "synthetic.safetensors")
        save_file(ckpt, path)
        conv = mc.load_from_standard_weights(path, "cpu")
    elif container == "memory":
        conv = mc.convert_state_dict(ckpt)
    else:
        path = str(tmp_path / "synthetic.ckpt")
        torch.save({"state_dict": ckpt, "global_step": 1} if container == "ckpt_state_dict" else ckpt, path)
        conv = mc.load_from_standard_weights(path, "cpu")
    check_against_golden(conv, gold)


def test_single_source_rules_alias_the_loaded_tensor(gold):
    ckpt = synthetic_checkpoint(gold)
    conv = mc.convert_state_dict(ckpt)
    assert conv["diffusion"]["final.conv.weight"] is ckpt["model.diffusion_model.out.2.weight"]
    assert conv["clip"]["embedding.position_embedding"] is \
        ckpt["cond_stage_model.transformer.text_model.embeddings.position_embedding.weight"]


def test_missing_key_is_a_keyerror_naming_it(gold):
    ckpt = synthetic_checkpoint(gold)
    del ckpt["first_stage_model.quant_conv.bias"]
    with pytest.raises(KeyError, match="first_stage_model.quant_conv.bias"):
        mc.convert_state_dict(ckpt)


def test_bad_safetensors_path_is_a_valueerror(tmp_path):
    with pytest.raises(ValueError, match="Error loading safetensors file"):
        mc.load_from_standard_weights(str(tmp_path / "nope.safetensors"), "cpu")


def test_real_shapes_round_trip_on_meta():
    """Full-size SD v1.5 shapes without the 4 GB: meta tensors with the reference modules' shapes ->
    CompVis layout -> back.  Shapes and keys must survive, the VAE attention weights must be 1x1 convs
    in the checkpoint layout and the fused projections must split three ways."""
    g = golden("state_dict_keys")
    sds = {}
    for m in mc.MODELS:
        sds[m] = {k: torch.empty([int(d) for d in s if d >= 0], device="meta")
                  for k, s in zip(g[m + "_keys"].tolist(), g[m + "_shapes"])}
    std = mc.to_standard_weights(sds)
    assert set(std) == set(mc.source_keys())
    assert tuple(std["first_stage_model.decoder.mid.attn_1.q.weight"].shape) == (512, 512, 1, 1)
    assert tuple(std["first_stage_model.encoder.mid.attn_1.proj_out.weight"].shape) == (512, 512, 1, 1)
    assert tuple(std["model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_k.weight"].shape) == (320, 320)
    assert tuple(std["cond_stage_model.transformer.text_model.encoder.layers.3.self_attn.v_proj.bias"].shape) == (768,)
    back = mc.convert_state_dict(std)
    for m in mc.MODELS:
        assert {k: tuple(v.shape) for k, v in back[m].items()} == {k: tuple(v.shape) for k, v in sds[m].items()}, m


def test_round_trip_values_small():
    """to_standard_weights is the exact inverse on values (hashed VAE encoder weights, 34 M params)."""
    from sd_amd.encoder import VAE_Encoder
    from sd_amd import loader
    enc = synth.load_hashed(VAE_Encoder(), "encoder.")
    sd = enc.state_dict()
    rules = [r for r in mc.conversion_rules() if r[0] == "encoder"]
    std = {}
    for _, dst, srcs, shape in rules:
        for s, part in zip(srcs, sd[dst].chunk(len(srcs), 0)):
            std[s] = part.reshape(part.shape + (1, 1)) if shape is not None else part
    back = {}
    for _, dst, srcs, shape in rules:
        t = std[srcs[0]] if len(srcs) == 1 else torch.cat([std[s] for s in srcs], 0)
        back[dst] = t if shape is None else t.reshape(shape)
    # the loader's meta-device construction adopts them strictly and bit-exactly
    m = loader._materialise(VAE_Encoder, back, torch.device("cpu"))
    got = m.state_dict()
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert not any(p.requires_grad for p in m.parameters())


def test_loader_refuses_cpu_and_side_vae(tmp_path):
    from sd_amd import loader
    from sd_amd._lib import SdamdError
    with pytest.raises(SdamdError):
        loader.preload_models_from_standard_weights(str(tmp_path / "x.ckpt"), device="cpu")
    if torch.cuda.is_available():
        with pytest.raises(NotImplementedError):
            loader.preload_models_from_standard_weights(str(tmp_path / "x.ckpt"), vae_checkpoint_path="v.pt")
