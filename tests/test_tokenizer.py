"""Text -> ids boundary.  No reference fixture pins it (the vocabulary is not shipped): parity
unpinned.  What is checked: a CLIP-style BPE built from user files flows through the same call the reference makes
(pipeline.py:51-53) into the [B, 77] int64 tensor the text encoder takes, on either transformers generation."""
import json

import pytest
import torch

import sd_amd  # noqa: F401
from sd_amd import pipeline, synth, tokenizer


def toy_files(tmp_path):
    chars = list("abcdefghijklmnopqrstuvwxyz")
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["ca"] = len(vocab)
    vocab["cat</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    v, m = tmp_path / "vocab.json", tmp_path / "merges.txt"
    v.write_text(json.dumps(vocab))
    m.write_text("#version: 0.2\nc a\nca t</w>\n")
    return vocab, str(v), str(m)


def test_clip_tokenizer_from_user_files(tmp_path):
    pytest.importorskip("transformers")
    vocab, v, m = toy_files(tmp_path)
    tok = tokenizer.load_clip_tokenizer(v, m)
    ids = pipeline._token_tensor(tok, ["a cat", ""], "cpu")
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (2, 77)
    bos, eos = vocab["<|startoftext|>"], vocab["<|endoftext|>"]
    assert ids[0, :4].tolist() == [bos, vocab["a</w>"], vocab["cat</w>"], eos] and (ids[0, 4:] == eos).all()
    assert ids[1, 0] == bos and (ids[1, 1:] == eos).all()      # the empty (unconditional) prompt: BOS + 76 x EOS


def test_missing_files_are_reported(tmp_path):
    with pytest.raises(FileNotFoundError, match="vocab.json"):
        tokenizer.load_clip_tokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))


def test_synthetic_tokenizer_layout():
    ids = pipeline._token_tensor(synth.SyntheticTokenizer(), ["synthetic:3", ""], "cpu")
    assert tuple(ids.shape) == (2, 77) and ids[0, 0] == synth.BOS and ids[0, -1] == synth.EOS
    assert ids[1].tolist() == synth.EMPTY_PROMPT_IDS
