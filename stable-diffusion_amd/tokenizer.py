"""Text -> token ids for ``generate(tokenizer=...)``.

The reference builds ``transformers.CLIPTokenizer("../data/vocab.json", merges_file="../data/merges.txt")``
in its notebook (test.ipynb cell 0) and only ever calls
``tokenizer.batch_encode_plus([text], padding="max_length", max_length=77).input_ids`` (pipeline.py:51-53,
57-59, 66-68).  The vocabulary files are not part of the reference repository (.gitignore:179) nor of this
one; ``load_clip_tokenizer`` builds the same object from files the user supplies.  Text -> ids is
third-party code on both sides: no reference fixture pins it (parity unpinned at this boundary), the
parity tests feed identical id tensors through ``synth.SyntheticTokenizer`` instead.
"""
import os


def load_clip_tokenizer(vocab_file, merges_file):
    for f in (vocab_file, merges_file):
        if not os.path.isfile(f):
            raise FileNotFoundError(f"CLIP tokenizer file not found: {f} (vocab.json / merges.txt of "
                                    "openai/clip-vit-large-patch14; supply them, nothing is downloaded)")
    import inspect
    from transformers import CLIPTokenizer
    if "vocab_file" in inspect.signature(CLIPTokenizer.__init__).parameters:   # transformers 4.x: the notebook's call
        return CLIPTokenizer(vocab_file, merges_file=merges_file)
    # transformers 5.x takes (vocab, merges), both file names or both in memory; the notebook's call no longer works
    return CLIPTokenizer(vocab=vocab_file, merges=merges_file)
