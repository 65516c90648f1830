"""Synthetic, bit-reproducible inputs for the SD-v1.5 hot path.

There is no network on the build or GPU boxes, so neither the v1.5 checkpoint nor the
CLIP vocabulary is available.  Everything the benches and parity tests
feed to the models is generated here from integer hashes:

* ``hash_tensor`` / ``hash_state_dict`` -- weights keyed by the reference's state_dict
  names (``loader.py:18-27`` loads them with ``strict=True``, so the names are the ABI).
  Values are exact dyadic rationals times one float32 scale: no libm, identical on any host.
* ``synthetic_token_ids`` -- token ids with the layout ``CLIPTokenizer`` would give
  (BOS 49406, body < 49406, EOS/pad 49407, length 77; ``pipeline.py:51-53``).
* ``SyntheticTokenizer`` -- duck-typed stand-in for the ``tokenizer`` kwarg of ``generate``.
"""
import zlib

import numpy as np
import torch

BOS, EOS, VOCAB, CTX_LEN = 49406, 49407, 49408, 77


def _lowbias32(x: np.ndarray) -> np.ndarray:
    """32-bit integer finaliser (wraps mod 2**32; numpy uint32 array arithmetic)."""
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def key_seed(key: str) -> int:
    return zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF


def hash_uniform(key: str, numel: int) -> np.ndarray:
    """float32[numel] of exact dyadic uniforms in [-1, 1): (top 24 hash bits - 2^23) * 2^-23."""
    salt = np.uint32((key_seed(key) * 0x9E3779B1) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = np.arange(numel, dtype=np.uint32)
        x ^= salt
        x = _lowbias32(x)
    v = (x >> np.uint32(8)).astype(np.int32) - np.int32(1 << 23)
    return v.astype(np.float32) * np.float32(2.0 ** -23)


def _scale_for(key: str, shape) -> tuple:
    """(scale, offset) per parameter kind; returns float32 scalars."""
    if len(shape) >= 2 and key.endswith(".weight"):
        fan_in = int(np.prod(shape[1:]))
        return np.float32(np.sqrt(np.float32(3.0) / np.float32(fan_in))), np.float32(0.0)
    if len(shape) == 1 and key.endswith(".weight"):      # GroupNorm / LayerNorm gain
        return np.float32(0.125), np.float32(1.0)
    if key.endswith(".bias"):
        return np.float32(0.0625), np.float32(0.0)
    return np.float32(0.0625), np.float32(0.0)             # e.g. embedding.position_embedding


def hash_tensor(key: str, shape) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    scale, offset = _scale_for(key, shape)
    v = hash_uniform(key, numel) * scale
    if offset != 0:
        v = v + offset
    return torch.from_numpy(v.reshape(shape))


def hash_state_dict(shapes: dict, prefix: str = "") -> dict:
    """shapes: {key: shape}.  ``prefix`` namespaces the hash (e.g. 'diffusion.')."""
    return {k: hash_tensor(prefix + k, s) for k, s in shapes.items()}


def shapes_of(module: torch.nn.Module) -> dict:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_hashed(module: torch.nn.Module, prefix: str) -> torch.nn.Module:
    """Fill ``module`` (reference, oracle or product: same key names) with hashed weights."""
    sd = hash_state_dict(shapes_of(module), prefix)
    module.load_state_dict(sd, strict=True)
    return module


def synthetic_token_ids(i: int) -> list:
    """Prompt ``i`` of the synthetic workload."""
    g = torch.Generator().manual_seed(1000 + i)
    body = 8 + (i % 32)
    ids = torch.randint(0, BOS, (body,), generator=g).tolist()
    return [BOS] + ids + [EOS] * (CTX_LEN - 1 - body)


EMPTY_PROMPT_IDS = [BOS] + [EOS] * (CTX_LEN - 1)


class _Encoding:
    def __init__(self, ids):
        self.input_ids = ids


class SyntheticTokenizer:
    """Maps 'synthetic:<i>' -> synthetic_token_ids(i), '' -> the empty prompt.

    Same call shape as the reference's use of ``CLIPTokenizer`` (``pipeline.py:51-53``):
    ``tok.batch_encode_plus([prompt], padding="max_length", max_length=77).input_ids``.
    """

    def batch_encode_plus(self, prompts, padding="max_length", max_length=CTX_LEN):
        assert max_length == CTX_LEN
        out = []
        for p in prompts:
            if p.startswith("synthetic:"):
                out.append(synthetic_token_ids(int(p.split(":", 1)[1])))
            elif p == "":
                out.append(list(EMPTY_PROMPT_IDS))
            else:  # deterministic fallback: hash the text into ids
                h = hash_uniform("prompt:" + p, 16)
                ids = ((h + 1.0) * 0.5 * (BOS - 1)).astype(np.int64).tolist()
                out.append([BOS] + ids + [EOS] * (CTX_LEN - 1 - len(ids)))
        return _Encoding(out)
