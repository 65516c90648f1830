"""CompVis-layout SD v1.x checkpoint -> the four state_dicts this package loads.

Same contract as the reference's ``model_converter.load_from_standard_weights``
(model_converter.py:3-1071): returns ``{"diffusion", "encoder", "decoder", "clip"}``, each a
dict keyed by the module parameter names that ``loader.py:18-27`` feeds to
``load_state_dict(strict=True)``.  The reference spells the mapping out as ~1,000 literal
assignments; here it is GENERATED from the block structure of the three networks
(``conversion_rules``): 996 renames (model_converter.py:28-1023), 32+2+2 q/k/v concatenations
(:1024-1069, dim 0) and the four VAE attention 1x1-conv -> Linear reshapes (:1040-1045).

A rule is ``(model, dst_key, src_keys, shape)``: one source key = alias of the loaded tensor (no
copy, like the reference), several = ``torch.cat(dim 0)``, ``shape`` = final ``reshape``.

Reading: ``.safetensors`` through ``safe_open`` (model_converter.py:5-17) -- but only the tensors
the rules name are materialised (EMA / optimizer entries of a full checkpoint are skipped); any
other extension through ``torch.load``.  The reference unpickles with ``weights_only=False``
(:19); this loader uses the restricted unpickler and needs ``trusted=True`` to do what the
reference does.
"""
import torch

UNET = "model.diffusion_model."
VAE = "first_stage_model."
TEXT = "cond_stage_model.transformer.text_model."
MODELS = ("diffusion", "encoder", "decoder", "clip")


class _Rules(list):
    def one(self, model, dst, src):
        self.append((model, dst, (src,), None))

    def wb(self, model, dst, src):
        """weight + bias of one layer"""
        self.one(model, dst + ".weight", src + ".weight")
        self.one(model, dst + ".bias", src + ".bias")

    def cat(self, model, dst, srcs, shape=None):
        self.append((model, dst, tuple(srcs), shape))


# ---------------------------------------------------------------------------------- UNet
def _unet_res(r, dst, src, skip):
    """UNET_ResidualBlock <- openaimodel ResBlock (model_converter.py:34-43, 112-113)."""
    r.wb("diffusion", dst + ".groupnorm_feature", src + ".in_layers.0")
    r.wb("diffusion", dst + ".conv_feature", src + ".in_layers.2")
    r.wb("diffusion", dst + ".linear_time", src + ".emb_layers.1")
    r.wb("diffusion", dst + ".groupnorm_merged", src + ".out_layers.0")
    r.wb("diffusion", dst + ".conv_merged", src + ".out_layers.3")
    if skip:
        r.wb("diffusion", dst + ".residual_layer", src + ".skip_connection")


def _unet_attn(r, dst, src):
    """UNET_AttentionBlock <- SpatialTransformer with one BasicTransformerBlock
    (model_converter.py:44-66; fused self-attention in_proj :1024-1039)."""
    tb = src + ".transformer_blocks.0"
    r.wb("diffusion", dst + ".groupnorm", src + ".norm")
    r.wb("diffusion", dst + ".conv_input", src + ".proj_in")
    r.wb("diffusion", dst + ".attention_1.out_proj", tb + ".attn1.to_out.0")
    r.wb("diffusion", dst + ".linear_geglu_1", tb + ".ff.net.0.proj")
    r.wb("diffusion", dst + ".linear_geglu_2", tb + ".ff.net.2")
    for p in "qkv":
        r.one("diffusion", f"{dst}.attention_2.{p}_proj.weight", f"{tb}.attn2.to_{p}.weight")
    r.wb("diffusion", dst + ".attention_2.out_proj", tb + ".attn2.to_out.0")
    for i in (1, 2, 3):
        r.wb("diffusion", f"{dst}.layernorm_{i}", f"{tb}.norm{i}")
    r.wb("diffusion", dst + ".conv_output", src + ".proj_out")
    r.cat("diffusion", dst + ".attention_1.in_proj.weight", [f"{tb}.attn1.to_{p}.weight" for p in "qkv"])


def _unet_rules(r):
    r.wb("diffusion", "time_embedding.linear_1", UNET + "time_embed.0")
    r.wb("diffusion", "time_embedding.linear_2", UNET + "time_embed.2")
    mults, n_res, attn_levels = (1, 2, 4, 4), 2, (0, 1, 2)
    # down path: input_blocks.i <-> unet.encoders.i
    r.wb("diffusion", "unet.encoders.0.0", UNET + "input_blocks.0.0")
    i, ch = 1, mults[0]
    for lvl, m in enumerate(mults):
        for _ in range(n_res):
            _unet_res(r, f"unet.encoders.{i}.0", f"{UNET}input_blocks.{i}.0", skip=(ch != m))
            ch = m
            if lvl in attn_levels:
                _unet_attn(r, f"unet.encoders.{i}.1", f"{UNET}input_blocks.{i}.1")
            i += 1
        if lvl != len(mults) - 1:   # stride-2 conv ("op" in the checkpoint)
            r.wb("diffusion", f"unet.encoders.{i}.0", f"{UNET}input_blocks.{i}.0.op")
            i += 1
    # middle
    _unet_res(r, "unet.bottleneck.0", UNET + "middle_block.0", skip=False)
    _unet_attn(r, "unet.bottleneck.1", UNET + "middle_block.1")
    _unet_res(r, "unet.bottleneck.2", UNET + "middle_block.2", skip=False)
    # up path: every residual block eats a skip concat, so all of them have a 1x1 skip conv
    i = 0
    for lvl in reversed(range(len(mults))):
        for j in range(n_res + 1):
            _unet_res(r, f"unet.decoders.{i}.0", f"{UNET}output_blocks.{i}.0", skip=True)
            k = 1
            if lvl in attn_levels:
                _unet_attn(r, f"unet.decoders.{i}.{k}", f"{UNET}output_blocks.{i}.{k}")
                k += 1
            if lvl != 0 and j == n_res:
                r.wb("diffusion", f"unet.decoders.{i}.{k}.conv", f"{UNET}output_blocks.{i}.{k}.conv")
            i += 1
    r.wb("diffusion", "final.groupnorm", UNET + "out.0")
    r.wb("diffusion", "final.conv", UNET + "out.2")


# ---------------------------------------------------------------------------------- VAE
def _vae_res(r, model, dst, src, shortcut):
    for a, b in (("groupnorm_1", "norm1"), ("conv_1", "conv1"), ("groupnorm_2", "norm2"), ("conv_2", "conv2")):
        r.wb(model, f"{dst}.{a}", f"{src}.{b}")
    if shortcut:
        r.wb(model, dst + ".residual_layer", src + ".nin_shortcut")


def _vae_attn(r, model, dst, src, ch=512):
    """1x1-conv q/k/v/proj_out of the checkpoint -> fused Linear in_proj / out_proj
    (model_converter.py:1040-1045: the reshapes are to the literal (1536, 512) / (512, 512))."""
    r.wb(model, dst + ".groupnorm", src + ".norm")
    r.one(model, dst + ".attention.out_proj.bias", src + ".proj_out.bias")
    r.cat(model, dst + ".attention.in_proj.weight", [f"{src}.{p}.weight" for p in "qkv"], (3 * ch, ch))
    r.cat(model, dst + ".attention.in_proj.bias", [f"{src}.{p}.bias" for p in "qkv"])
    r.cat(model, dst + ".attention.out_proj.weight", [src + ".proj_out.weight"], (ch, ch))


def _vae_rules(r):
    mults = (1, 2, 4, 4)
    # encoder: nn.Sequential indices 0..18 (encoder.py:8-58)
    E = VAE + "encoder."
    r.wb("encoder", "0", E + "conv_in")
    i, ch = 1, mults[0]
    for lvl, m in enumerate(mults):
        for b in range(2):
            _vae_res(r, "encoder", str(i), f"{E}down.{lvl}.block.{b}", shortcut=(ch != m))
            ch = m
            i += 1
        if lvl != len(mults) - 1:
            r.wb("encoder", str(i), f"{E}down.{lvl}.downsample.conv")
            i += 1
    _vae_res(r, "encoder", str(i), E + "mid.block_1", False)
    _vae_attn(r, "encoder", str(i + 1), E + "mid.attn_1")
    _vae_res(r, "encoder", str(i + 2), E + "mid.block_2", False)
    r.wb("encoder", str(i + 3), E + "norm_out")          # i+4 is the SiLU
    r.wb("encoder", str(i + 5), E + "conv_out")
    r.wb("encoder", str(i + 6), VAE + "quant_conv")
    # decoder: nn.Sequential indices 0..25 (decoder.py:75-124); checkpoint "up" levels count from
    # the output end
    D = VAE + "decoder."
    r.wb("decoder", "0", VAE + "post_quant_conv")
    r.wb("decoder", "1", D + "conv_in")
    _vae_res(r, "decoder", "2", D + "mid.block_1", False)
    _vae_attn(r, "decoder", "3", D + "mid.attn_1")
    _vae_res(r, "decoder", "4", D + "mid.block_2", False)
    i, ch = 5, mults[-1]
    for lvl in reversed(range(len(mults))):
        for b in range(3):
            _vae_res(r, "decoder", str(i), f"{D}up.{lvl}.block.{b}", shortcut=(ch != mults[lvl]))
            ch = mults[lvl]
            i += 1
        if lvl != 0:
            r.wb("decoder", str(i + 1), f"{D}up.{lvl}.upsample.conv")   # i is the nn.Upsample
            i += 2
    r.wb("decoder", str(i), D + "norm_out")               # i+1 is the SiLU
    r.wb("decoder", str(i + 2), D + "conv_out")


# ---------------------------------------------------------------------------------- CLIP
def _clip_rules(r, n_layers=12):
    r.one("clip", "embedding.token_embedding.weight", TEXT + "embeddings.token_embedding.weight")
    r.one("clip", "embedding.position_embedding", TEXT + "embeddings.position_embedding.weight")
    for i in range(n_layers):
        src, dst = f"{TEXT}encoder.layers.{i}", f"layers.{i}"
        r.wb("clip", dst + ".attention.out_proj", src + ".self_attn.out_proj")
        r.wb("clip", dst + ".layernorm_1", src + ".layer_norm1")
        r.wb("clip", dst + ".linear_1", src + ".mlp.fc1")
        r.wb("clip", dst + ".linear_2", src + ".mlp.fc2")
        r.wb("clip", dst + ".layernorm_2", src + ".layer_norm2")
        for kind in ("weight", "bias"):
            r.cat("clip", f"{dst}.attention.in_proj.{kind}", [f"{src}.self_attn.{p}_proj.{kind}" for p in "qkv"])
    r.wb("clip", "layernorm", TEXT + "final_layer_norm")


_RULES = None


def conversion_rules():
    """[(model, dst_key, src_keys, reshape_or_None)] for SD v1.x (1,042 destination keys)."""
    global _RULES
    if _RULES is None:
        r = _Rules()
        _unet_rules(r)
        _vae_rules(r)
        _clip_rules(r)
        _RULES = tuple(r)
    return _RULES


def source_keys():
    """Every checkpoint key the conversion reads (in rule order, no duplicates)."""
    seen = {}
    for _, _, srcs, _ in conversion_rules():
        for s in srcs:
            seen.setdefault(s)
    return list(seen)


def convert_state_dict(original_model):
    """Apply the rules to an in-memory mapping ``{checkpoint key: tensor}``.  A key the rules need
    and the mapping lacks raises ``KeyError`` naming it (the reference's behaviour)."""
    converted = {m: {} for m in MODELS}
    for model, dst, srcs, shape in conversion_rules():
        t = original_model[srcs[0]] if len(srcs) == 1 else torch.cat([original_model[s] for s in srcs], 0)
        converted[model][dst] = t if shape is None else t.reshape(shape)
    return converted


def to_standard_weights(state_dicts):
    """Inverse of ``convert_state_dict``: the four state_dicts -> one CompVis-layout mapping
    (fused projections split back, VAE attention weights back to 1x1-conv shape)."""
    out = {}
    for model, dst, srcs, shape in conversion_rules():
        t = state_dicts[model][dst]
        for s, part in zip(srcs, t.chunk(len(srcs), 0)):
            out[s] = part.reshape(part.shape + (1, 1)) if shape is not None else part
    return out


def _read_safetensors(path, device, keys):
    try:
        from safetensors import safe_open
    except ImportError:
        raise ImportError("Please install safetensors: pip install safetensors")
    try:
        with safe_open(path, framework="pt", device=str(device)) as f:
            have = set(f.keys())
            return {k: f.get_tensor(k) for k in keys if k in have}
    except Exception as e:   # same message shape as model_converter.py:16-17
        raise ValueError(f"Error loading safetensors file: {str(e)}. Make sure the file path is correct.")


def load_from_standard_weights(input_file: str, device: str, trusted: bool = False) -> dict:
    if str(input_file).endswith(".safetensors"):
        original_model = _read_safetensors(input_file, device, source_keys())
    else:
        checkpoint = torch.load(input_file, map_location=device, weights_only=not trusted)
        original_model = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    return convert_state_dict(original_model)
