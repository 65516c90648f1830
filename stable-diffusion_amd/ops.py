"""Host-side operators: torch tensors in, one C-ABI launch each (include/sdamd.h).

PyTorch is only the tensor host here (device memory, streams).  Every function checks the
shapes/dtypes the kernels assume before launching, uses torch's current HIP stream (so the
launches are capturable into a hipGraph) and never falls back to torch eager math.

Activation layout is NHWC / token-major: [B, H, W, C] or [B, S, C], float32 or bfloat16.
"""
import ctypes

import torch

from . import _lib

F32, BF16 = 0, 1
ACT_NONE, ACT_QUICK_GELU, ACT_SILU, ACT_GEGLU = 0, 1, 2, 3


def _code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"sd_amd kernels compute in float32 or bfloat16, got {dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.SdamdError("sd_amd ops need tensors on the HIP device (no CPU fallback)")


def _p(t):
    return 0 if t is None else t.data_ptr()


class LaunchTimer:
    """Optional per-launch timing with HIP events recorded on the launch stream (used by
    bench.py's roofline leg; None in normal operation).  records: (kind, work, start, end, shape, bytes) with
    ``work`` = algorithmic FLOPs of the REFERENCE's op (MFMA kernels: 2*M*N*K of the nn.Linear / nn.Conv2d the
    launch stands for -- the hi|lo operand doubling and the 4/9-MAC upsample form are not credited) or bytes
    (bandwidth kernels), ``bytes`` = compulsory HBM bytes of the launch (operands + weights + result + residual,
    each once)."""

    def __init__(self):
        self.records = []

    def _fold(self, key_of):
        torch.cuda.synchronize()
        out = {}
        for rec in self.records:
            kind, work, s, e, shape, nbytes = rec
            ent = out.setdefault(key_of(kind, shape), {"launches": 0, "ms": 0.0, "work": 0.0, "bytes": 0.0, "times_ms": []})
            dt = s.elapsed_time(e)
            ent["launches"] += 1
            ent["ms"] += dt
            ent["work"] += work
            ent["bytes"] += nbytes
            ent["times_ms"].append(dt)
        return out

    def summary(self):
        return self._fold(lambda kind, shape: kind)

    def by_shape(self):
        """{(kind, shape): {"launches", "ms", "work", "bytes"}} -- per problem size, for tuning."""
        return self._fold(lambda kind, shape: (kind, shape))


PROFILER = None


def _timed(kind, work, launch, shape="", nbytes=0.0):
    prof = PROFILER
    if prof is None:
        return launch()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = launch()
    e.record()
    prof.records.append((kind() if callable(kind) else kind, float(work), s, e, shape() if callable(shape) else shape, float(nbytes)))
    return r


def _tile_tag(M, N, K, dtype, ws, batch=1, act=0, f32_residual=False, conv=False):
    """Profiler label of the kernel sdamd_gemm / sdamd_conv3x3 will pick: tile, split-K factor, and ``_rp`` / ``_geglu``
    for the residual-prefetch (gemm2.hip launch2: f32 residual, no split, 128x160 or dense 128x320 tile) and GeGLU-only
    epilogue instantiations."""
    t = _lib.lib().sdamd_gemm_tile(M, N, K, batch, _code(dtype), ws.numel(), int(act))
    t, s = t % 1000000, t // 1000000 + 1
    rp = f32_residual and s == 1 and gemm_supports_act(ACT_GEGLU, torch.bfloat16, 256) and (t in (64160, 128160) or (t == 128320 and not conv))
    return f"{t // 1000}x{t % 1000}" + (f"_splitk{s}" if s > 1 else "_rp" if rp else "_geglu" if int(act) == ACT_GEGLU and t != 0 else "")


_gemm_ws = {}
_gemm_ws_retired = []     # outgrown buffers stay alive: an earlier captured graph may still point at them
GEMM_WS_BYTES = 64 << 20
PLAN_SAMPLES = 1          # > 1: deterministic launch plans for batches of that many samples (plan_samples below)


class plan_samples:
    """``with ops.plan_samples(n):`` -- the launches inside span n independent samples and are planned (tile, split-K,
    kernel family, GroupNorm partial blocks, fusion thresholds) from ONE sample's rows (sdamd_set_plan_samples), so each
    sample's result is bit-identical to its result in any other batch size run the same way.  n = 0 / None: no change.
    The split-K workspace grows to n x GEMM_WS_BYTES (same split limits per sample)."""

    def __init__(self, n):
        self.n = int(n) if n else 0

    def __enter__(self):
        global PLAN_SAMPLES
        self.old = PLAN_SAMPLES
        if self.n >= 1:
            PLAN_SAMPLES = self.n
            _lib.lib().sdamd_set_plan_samples(self.n)
        return self

    def __exit__(self, *exc):
        global PLAN_SAMPLES
        PLAN_SAMPLES = self.old
        _lib.lib().sdamd_set_plan_samples(self.old)


def _plan_rows(M):
    return M // PLAN_SAMPLES if PLAN_SAMPLES > 1 and M % PLAN_SAMPLES == 0 else M


def _gemm_workspace(device):
    """Per-device scratch for split-K partial sums (sdamd_gemm / sdamd_conv3x3 `ws`), allocated once and never freed
    or replaced, so a captured hipGraph that holds its address stays valid.  Use is stream-ordered: the package
    launches on one stream per process (torch's current stream, or the capture stream while a graph is recorded);
    launches that overlap in time on two streams of one device must not share it."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    ws = _gemm_ws.get(key)
    need = GEMM_WS_BYTES * max(1, PLAN_SAMPLES)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.SdamdError("split-K workspace (re)allocation under graph capture: run the ops once "
                                  "(or ops.ensure_workspaces) before capturing")
        if ws is not None:
            _gemm_ws_retired.append(ws)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _gemm_ws[key] = ws
    if ws.numel() != need:
        ws = ws[:need]       # the planner's split limit must not depend on what an earlier, larger request left behind
    return ws


def _dt(dtype):
    return "bf16" if dtype == torch.bfloat16 else "f32"


_gn_ws = {}
_gn_ws_retired = []      # outgrown buffers stay alive: an earlier captured graph may still point at them
GN_WS_MIN_BYTES = 8 << 20   # 256 samples


def _groupnorm_ws(batch, device):
    """Per-device GroupNorm partial-statistics scratch (same lifetime / stream rules as _gemm_workspace)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    need = _lib.lib().sdamd_groupnorm_ws_bytes(batch)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.SdamdError("GroupNorm workspace (re)allocation under graph capture: run the ops once "
                                  "(or ops.ensure_workspaces) before capturing")
        if ws is not None:
            _gn_ws_retired.append(ws)
        ws = torch.empty(max(need, GN_WS_MIN_BYTES), dtype=torch.uint8, device=device)
        _gn_ws[key] = ws
    return ws


def ensure_workspaces(device, batch=1):
    """Allocate the per-device scratch buffers ahead of a graph capture."""
    _gemm_workspace(device)
    _groupnorm_ws(batch, device)


# ------------------------------------------------------------------ sampler
def cfg_ddpm_step(latents, eps, noise, coef, step, cfg_scale, do_cfg=True, step_ptr=None, deterministic=False):
    """In-place on ``latents`` [B,4,h,w] f32; eps [2B,4,h,w] f32 ([pos..., neg...]).  ``cfg_scale=None``: the kernel
    reads the guidance scale from column 7 of the step's coefficient row.  ``noise`` may only be None
    for a schedule the caller has checked to be noise-free (``deterministic=True``: every coef[:, 5] == 0); the
    kernel reads a null noise pointer as "no noise".  A device-side index (``step_ptr``) outside the schedule makes
    the launch a no-op."""
    _dev(latents, eps, coef)
    B = latents.shape[0]
    n = latents[0].numel()
    assert latents.dtype == torch.float32 and eps.dtype == torch.float32 and coef.dtype == torch.float32
    assert latents.is_contiguous() and eps.is_contiguous() and coef.is_contiguous()
    assert eps.numel() == (2 if do_cfg else 1) * B * n, (eps.shape, latents.shape)
    assert coef.dim() == 2 and coef.shape[1] == 8
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.numel() == coef.shape[0] * B * n
    elif not deterministic:
        raise _lib.SdamdError("cfg_ddpm_step: noise=None needs deterministic=True (a coefficient table with has_noise == 0 "
                              "in every row); a stochastic sampler must pass its pre-drawn noise")
    if step_ptr is not None:
        assert step_ptr.dtype == torch.int32 and step_ptr.is_cuda
    elif not 0 <= step < coef.shape[0]:
        raise IndexError(f"cfg_ddpm_step: step {step} outside the {coef.shape[0]}-step schedule")
    # algorithmic bytes: 5 x 4*h*w x 4 B per image-step (eps pos/neg, noise, latents in + out)
    _timed("cfg_ddpm_step", 5.0 * B * n * 4,
           lambda: _lib.check(_lib.lib().sdamd_cfg_ddpm_step(_p(latents), _p(eps), _p(noise), _p(coef), _p(step_ptr), int(step),
                                                             coef.shape[0], float("nan") if cfg_scale is None else float(cfg_scale),
                                                             int(do_cfg), B, n, _stream())),
           f"B{B} n{n}")
    return latents


def counter_add(counter, delta=1):
    assert counter.dtype == torch.int32 and counter.is_cuda
    _lib.check(_lib.lib().sdamd_counter_add(_p(counter), int(delta), _stream()))


def add_noise(x, noise, sqrt_abar, sqrt_one_minus_abar):
    _dev(x, noise)
    assert x.dtype == torch.float32 and noise.dtype == torch.float32 and x.shape == noise.shape
    x, noise = x.contiguous(), noise.contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().sdamd_add_noise(_p(out), _p(x), _p(noise), float(sqrt_abar), float(sqrt_one_minus_abar),
                                          x.numel(), _stream()))
    return out


def gather_row(dst, table, step=0, step_ptr=None):
    _dev(dst, table)
    assert dst.dtype == torch.float32 and table.dtype == torch.float32 and table.dim() == 2
    assert dst.numel() == table.shape[1] and dst.is_contiguous() and table.is_contiguous()
    if step_ptr is None and not 0 <= step < table.shape[0]:
        raise IndexError(f"gather_row: row {step} outside the {table.shape[0]}-row table")
    _lib.check(_lib.lib().sdamd_gather_row(_p(dst), _p(table), _p(step_ptr), int(step), table.shape[0], table.shape[1],
                                           _stream()))
    return dst


# ------------------------------------------------------------------ norms
def groupnorm(x, gamma, beta, eps, silu, out_dtype=None, x2=None):
    """x [B, ..., C] NHWC; 32 groups.  ``out_dtype`` (default: x's) may be bfloat16 for f32 input.
    ``x2``: normalise the channel concat [x | x2] without materialising it."""
    _dev(x, gamma, beta)
    assert x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    B, C1 = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        _dev(x2)
        assert x2.is_contiguous() and x2.dtype == x.dtype and x2.shape[:-1] == x.shape[:-1]
        C2 = x2.shape[-1]
    C = C1 + C2
    assert gamma.numel() == C and beta.numel() == C
    out_dtype = x.dtype if out_dtype is None else out_dtype
    y = torch.empty(x.shape[:-1] + (C,), dtype=out_dtype, device=x.device)
    ws = _groupnorm_ws(B, x.device)
    L = _lib.lib()
    # statistics left behind by the producing GEMM / conv epilogue (gemm / conv3x3 want_stats): no pass over x for them
    st1 = _stats_of(x)
    st2 = None if x2 is None else _stats_of(x2)
    pre = USE_PRODUCER_STATS and st1 is not None and (x2 is None or st2 is not None) and HW % 64 == 0
    if pre:
        assert st1.shape == (B * HW // 64, C1, 2) and (st2 is None or st2.shape == (B * HW // 64, C2, 2))
        call = lambda: _lib.check(L.sdamd_groupnorm_nhwc_pre(_p(x), _p(x2), C1, C2, _p(y), _p(gamma), _p(beta), _p(st1), _p(st2),
                                                             _p(ws), B, HW, float(eps), int(silu), _code(x.dtype),
                                                             _code(out_dtype), _stream()))
    elif x2 is None:
        call = lambda: _lib.check(L.sdamd_groupnorm_nhwc(_p(x), _p(y), _p(gamma), _p(beta), _p(ws), B, HW, C, float(eps),
                                                         int(silu), _code(x.dtype), _code(out_dtype), _stream()))
    else:
        call = lambda: _lib.check(L.sdamd_groupnorm_nhwc_cat(_p(x), _p(x2), C1, C2, _p(y), _p(gamma), _p(beta), _p(ws), B, HW,
                                                             float(eps), int(silu), _code(x.dtype), _code(out_dtype), _stream()))
    _timed(f"groupnorm_{_dt(x.dtype)}" + ("_pre" if pre else ""), B * HW * C * (x.element_size() + y.element_size()),   # 1 read + 1 write per element
           call, f"B{B} HW{HW} C{C}" + (" cat" if x2 is not None else ""))
    return y


USE_PRODUCER_STATS = True     # tests / A-B switch: False makes every GroupNorm compute its own statistics
# Below ~8 M elements the statistics pass costs no more than the finalize launch that replaces it (measured at UNet
# batch 16, experiments/ab_gnpre.py: 64x64x320 41.9 -> 28.1 us, 32x32x640 26.9 -> 19.4 us, 16x16x1280 15.9 -> 16.0 us)
PRODUCER_STATS_MIN_ELEMS = 8 << 20


def layernorm(x, gamma, beta, eps=1e-5, out_dtype=None):
    _dev(x, gamma, beta)
    assert x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    C = x.shape[-1]
    assert gamma.numel() == C and beta.numel() == C
    out_dtype = x.dtype if out_dtype is None else out_dtype
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _timed(f"layernorm_{_dt(x.dtype)}", x.numel() * (x.element_size() + y.element_size()),
           lambda: _lib.check(_lib.lib().sdamd_layernorm(_p(x), _p(y), _p(gamma), _p(beta), x.numel() // C, C, float(eps),
                                                         _code(x.dtype), _code(out_dtype), _stream())),
           f"rows{x.numel() // C} C{C}")
    return y


# ------------------------------------------------------------------ elementwise
def geglu(x):
    """x [..., 2F] (value | gate) -> [..., F]."""
    _dev(x)
    assert x.is_contiguous() and x.shape[-1] % 2 == 0
    Fh = x.shape[-1] // 2
    out = torch.empty(x.shape[:-1] + (Fh,), dtype=x.dtype, device=x.device)
    _timed(f"geglu_{_dt(x.dtype)}", (x.numel() + out.numel()) * x.element_size(),
           lambda: _lib.check(_lib.lib().sdamd_geglu(_p(x), _p(out), x.numel() // (2 * Fh), Fh, _code(x.dtype), _stream())),
           f"rows{x.numel() // (2 * Fh)} F{Fh}")
    return out


def clip_embed(ids, table, pos):
    _dev(ids, table, pos)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and table.dtype == pos.dtype
    B, S = ids.shape
    D = table.shape[1]
    assert pos.shape == (S, D)
    out = torch.empty((B, S, D), dtype=table.dtype, device=table.device)
    _lib.check(_lib.lib().sdamd_clip_embed(_p(ids), _p(table), _p(pos), _p(out), B * S, S, D, table.shape[0],
                                           _code(table.dtype), _stream()))
    return out


def concat_channels(a, b):
    _dev(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[:-1] == b.shape[:-1]
    C1, C2 = a.shape[-1], b.shape[-1]
    out = torch.empty(a.shape[:-1] + (C1 + C2,), dtype=a.dtype, device=a.device)
    _lib.check(_lib.lib().sdamd_concat_nhwc(_p(a), _p(b), _p(out), a.numel() // C1, C1, C2, _code(a.dtype), _stream()))
    return out


def add(a, b):
    _dev(a, b)
    assert a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    out = torch.empty_like(a)
    _lib.check(_lib.lib().sdamd_add(_p(a), _p(b), _p(out), a.numel(), _code(a.dtype), _stream()))
    return out


def nchw_to_nhwc(x, dtype):
    """[B,C,H,W] f32 -> [B,H,W,C] dtype."""
    _dev(x)
    x = x.contiguous()
    assert x.dtype == torch.float32 and x.dim() == 4
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=dtype, device=x.device)
    _lib.check(_lib.lib().sdamd_nchw_to_nhwc(_p(x), _p(out), B, C, H * W, _code(dtype), _stream()))
    return out


def nhwc_to_nchw(x):
    """[B,H,W,C] dtype -> [B,C,H,W] f32."""
    _dev(x)
    assert x.is_contiguous() and x.dim() == 4
    B, H, W, C = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sdamd_nhwc_to_nchw(_p(x), _p(out), B, C, H * W, _code(x.dtype), _stream()))
    return out


def cast(x, dtype):
    _dev(x)
    x = x.contiguous()
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.check(_lib.lib().sdamd_cast(_p(x), _p(out), x.numel(), _code(x.dtype), _code(dtype), _stream()))
    return out


def split_hilo(x, x2=None):
    """f32 [..., K] -> bf16 [..., 2K] = [bf16(x) | bf16(x - bf16(x))] (pair with weights [W | W]).
    ``x2``: split the column concat [x | x2] without materialising it."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    K1 = x.shape[-1]
    rows = x.numel() // K1
    if x2 is None:
        out = torch.empty(x.shape[:-1] + (2 * K1,), dtype=torch.bfloat16, device=x.device)
        _timed("split_hilo", x.numel() * 8,
               lambda: _lib.check(_lib.lib().sdamd_split_hilo(_p(x), _p(out), rows, K1, _stream())), f"rows{rows} K{K1}")
        return out
    _dev(x2)
    assert x2.dtype == torch.float32 and x2.is_contiguous() and x2.shape[:-1] == x.shape[:-1]
    K2 = x2.shape[-1]
    out = torch.empty(x.shape[:-1] + (2 * (K1 + K2),), dtype=torch.bfloat16, device=x.device)
    _timed("split_hilo", rows * (K1 + K2) * 8,
           lambda: _lib.check(_lib.lib().sdamd_split_hilo_cat(_p(x), _p(x2), _p(out), rows, K1, K2, _stream())),
           f"rows{rows} K{K1}+{K2}")
    return out


def silu(x):
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().sdamd_silu(_p(x), _p(out), x.numel(), _stream()))
    return out


def image_to_uint8(x):
    """[B,3,H,W] f32 in (-1,1) -> [B,H,W,3] uint8 (truncating, pipeline.py:158-160)."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == 3
    B, _, H, W = x.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().sdamd_image_to_uint8(_p(x), _p(out), B, H * W, _stream()))
    return out


def uint8_to_image(x):
    """[B,H,W,3] uint8 -> [B,3,H,W] f32 in (-1,1) (pipeline.py:99-104)."""
    _dev(x)
    assert x.dtype == torch.uint8 and x.is_contiguous() and x.shape[-1] == 3
    B, H, W, _ = x.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sdamd_uint8_to_image(_p(x), _p(out), B, H * W, _stream()))
    return out


def vae_post_quant(lat, w, b):
    _dev(lat, w, b)
    assert lat.dtype == torch.float32 and lat.is_contiguous() and lat.shape[1] == 4 and w.numel() == 16 and b.numel() == 4
    out = torch.empty_like(lat)
    _lib.check(_lib.lib().sdamd_vae_post_quant(_p(lat), _p(out), _p(w), _p(b), lat.shape[0], lat.shape[2] * lat.shape[3],
                                               _stream()))
    return out


def vae_sample(moments, noise, w, b):
    _dev(moments, noise, w, b)
    assert moments.dtype == torch.float32 and moments.is_contiguous() and moments.shape[1] == 8
    noise = noise.contiguous()
    B, _, H, W = moments.shape
    assert noise.shape == (B, 4, H, W) and noise.dtype == torch.float32 and w.numel() == 64 and b.numel() == 8
    out = torch.empty_like(noise)
    _lib.check(_lib.lib().sdamd_vae_sample(_p(moments), _p(noise), _p(out), _p(w), _p(b), B, H * W, _stream()))
    return out


# ------------------------------------------------------------------ MFMA GEMM / conv / attention
def gemm_hilo_fused(M, N, K, dtype):
    """True when sdamd_gemm can emit its result directly as the bf16 hi|lo pair (out_f32 = 2)."""
    if dtype != torch.bfloat16 or not gemm_supports_act(ACT_GEGLU, dtype, 256):    # 4-wave kernels forced
        return False
    tile = _lib.lib().sdamd_gemm_tile(M, N, K, 1, _code(dtype), GEMM_WS_BYTES * max(1, PLAN_SAMPLES), 0) % 1000000
    return tile in (64160, 128160, 128320, 256256, 256320, 256128)      # the LDS-DMA tiles (gemm2.hip); the register-staged ones have no such epilogue


def gemm(a, w, bias=None, residual=None, act=ACT_NONE, out_f32=False, out_hilo=False, flops_k=None, want_stats=False, out=None):
    """a [..., K] (last dim contiguous, uniform row stride), w [N, K] -> [..., N].  The result (and the
    residual, which must have the result's dtype) is f32 when ``out_f32``, else a's dtype.
    ``out_hilo``: f32 residual, result bf16 [..., 2N] = [bf16(v) | bf16(v - bf16(v))] (see split_hilo).
    ``flops_k``: the K of the reference's op when it differs from the executed one (hi|lo operands run 2K), for the
    profiler's algorithmic-FLOP credit only.
    ``want_stats``: the result feeds a GroupNorm: ask the epilogue for per-column partial statistics; when the launch
    produced them they ride on the result as ``out._sd_colstats`` ([ceil(M/64), N, 2] f32) and ``groupnorm`` uses them."""
    _dev(a, w, bias, residual)
    K = a.shape[-1]
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous() and w.dtype == a.dtype, (a.shape, w.shape, a.dtype, w.dtype)
    assert a.stride(-1) == 1
    lead = a.shape[:-1]
    M = 1
    for s in lead:
        M *= s
    if a.dim() == 2:
        lda = a.stride(0) if M > 1 else K
    else:
        assert a.is_contiguous(), "gemm: batched inputs must be contiguous"
        lda = K
    out_dtype = torch.float32 if out_f32 else a.dtype
    n_out = N // 2 if act == ACT_GEGLU else N     # GEGLU: w rows are value|gate tiles, see geglu_pack
    res_dtype = out_dtype
    if out_hilo:
        assert a.dtype == torch.bfloat16 and act == ACT_NONE and not out_f32
        res_dtype, n_out, out_f32 = torch.float32, 2 * N, 2
    if out is None:
        out = torch.empty(lead + (n_out,), dtype=out_dtype, device=a.device)
    else:   # caller's buffer (e.g. one half of a batch): same shape, dtype, dense
        assert out.dtype == out_dtype and out.is_contiguous() and out.numel() == M * n_out and out.shape[-1] == n_out and out.device == a.device
    ldr = 0
    if residual is not None:
        assert residual.dtype == res_dtype and residual.is_contiguous() and residual.numel() == M * (N if out_hilo else n_out), \
            (residual.dtype, res_dtype, residual.shape, M, N)
        ldr = N
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    ws = _gemm_workspace(a.device)
    nbytes = (M * K + N * K) * a.element_size() + out.numel() * out.element_size() + (0 if residual is None else residual.numel() * residual.element_size())
    stats, wrote = _stats_request(want_stats and act != ACT_GEGLU and not out_hilo, M, N, a.device)
    w3 = weight_planes(w)
    _timed(lambda: f"gemm_{_dt(a.dtype)}_{_tile_tag(M, N, K, a.dtype, ws, 1, act, residual is not None and res_dtype == torch.float32)}",
           2.0 * M * N * (K if flops_k is None else flops_k),
           lambda: _lib.check(_lib.lib().sdamd_gemm(_p(a), _p(w), _p(out), _p(bias), _p(residual), M, N, K, lda, K, n_out, ldr, 1,
                                                    0, 0, 0, 0, int(act), _code(a.dtype), int(out_f32), _p(ws), ws.numel(),
                                                    _p(stats), wrote, _p(w3), _stream())),
           lambda: f"M{M} N{N} K{K}{' +res' if residual is not None else ''}{' hilo' if out_f32 == 2 else ' f32out' if out_f32 else ''}{' geglu' if act == ACT_GEGLU else ''}",
           nbytes)
    _stats_attach(out, stats, wrote)
    return out


FUSE_LAYERNORM = True          # tests / A-B switch: False keeps GEMM and LayerNorm separate launches
GEMM_LN_MIN_ROWS = 32768       # below this the 128x320 tile leaves most CUs idle (M / 128 workgroups) and two launches are no slower


def gemm_ln_fused(M, N, dtype):
    """True when ``gemm_ln`` runs as ONE launch (sdamd_gemm_ln): bf16 operands, 320-wide rows, enough rows to fill the chip."""
    return FUSE_LAYERNORM and dtype == torch.bfloat16 and N == 320 and _plan_rows(M) >= GEMM_LN_MIN_ROWS and gemm_supports_act(ACT_GEGLU, dtype, 256)


def gemm_ln(a, w, bias, residual, gamma, beta, eps, out=None, ln_out=None):
    """(t, n): t [..., N] f32 = a @ w^T + bias (+ residual f32), n [..., N] bf16 = LayerNorm(t) * gamma + beta -- one launch
    whose epilogue normalises the rows it just stored (N == 320 on the 128x320 tile), instead of a LayerNorm launch that
    reads the f32 trunk back (diffusion.py:99-100, 110-117, 119-126).  Callers check ``gemm_ln_fused`` first."""
    _dev(a, w, bias, residual, gamma, beta)
    K, N = a.shape[-1], w.shape[0]
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.shape[1] == K and w.is_contiguous() and a.is_contiguous()
    lead = a.shape[:-1]
    M = a.numel() // K
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == N and beta.numel() == N
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == N)
    if out is None:
        out = torch.empty(lead + (N,), dtype=torch.float32, device=a.device)
    if ln_out is None:
        ln_out = torch.empty(lead + (N,), dtype=torch.bfloat16, device=a.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == M * N
    assert ln_out.dtype == torch.bfloat16 and ln_out.is_contiguous() and ln_out.numel() == M * N
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.numel() == M * N
    nbytes = (M * K + N * K) * 2 + M * N * (4 + 2) + (0 if residual is None else M * N * 4)
    _timed("gemm_bf16_128x320_ln", 2.0 * M * N * K,
           lambda: _lib.check(_lib.lib().sdamd_gemm_ln(_p(a), _p(w), _p(out), _p(bias), _p(residual), _p(ln_out), _p(gamma), _p(beta),
                                                       float(eps), M, N, K, K, _stream())),
           f"M{M} N{N} K{K}{' +res' if residual is not None else ''} f32out +ln", nbytes)
    return out, ln_out


def _stats_request(want, M, N, device):
    """(stats buffer, ctypes flag) for a launch that should leave GroupNorm column statistics, or (None, None)."""
    if not want or not USE_PRODUCER_STATS or _plan_rows(M) * N < PRODUCER_STATS_MIN_ELEMS:
        return None, None
    return torch.empty(((M + 63) // 64, N, 2), dtype=torch.float32, device=device), ctypes.byref(ctypes.c_int(0))


def _stats_attach(out, stats, wrote):
    """Invariant: the statistics describe ``out`` as the launch stored it.  The version the tensor had then rides along
    and ``groupnorm`` drops statistics whose tensor has since been written in place (a write through ``.data`` or
    another view of the storage does not bump the version: nothing in the package writes a producer's result)."""
    if stats is not None and wrote._obj.value:
        out._sd_colstats = stats
        out._sd_colstats_version = out._version


def _stats_of(x):
    st = getattr(x, "_sd_colstats", None)
    if st is not None and getattr(x, "_sd_colstats_version", None) != x._version:
        return None
    return st


F32_MODES = {"exact": 0, "bf16x3": 1}
USE_WEIGHT_PLANES = True      # tests / A-B switch: False makes the bf16x3 kernels split the weight fragments in registers
_W3 = {}                      # data_ptr -> (weakref to the f32 weight, its version, shape, bf16 planes)


def weight_planes(w):
    """The bf16x3 planes of a contiguous f32 weight matrix [..., K] (sdamd_split3_weights: [rows][K/32][hi|mid|lo][32]
    bf16), made once per weight tensor and kept while that tensor lives; None when the mode / shape has no use for them.
    Never made under graph capture (the warm-up run before a capture makes them)."""
    if not USE_WEIGHT_PLANES or w.dtype != torch.float32 or not w.is_contiguous() or w.shape[-1] % 32 or get_f32_mode() != "bf16x3":
        return None
    key = w.data_ptr()
    ent = _W3.get(key)
    if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == tuple(w.shape):
        return ent[3]
    if torch.cuda.is_current_stream_capturing():
        raise _lib.SdamdError("weight planes requested for the first time under graph capture: run the step once before capturing")
    K = w.shape[-1]
    rows = w.numel() // K
    planes = torch.empty((rows, 3 * K), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().sdamd_split3_weights(_p(w), _p(planes), rows, K, K, _stream()))
    import weakref
    _W3[key] = (weakref.ref(w, lambda _r, k=key: _W3.pop(k, None)), w._version, tuple(w.shape), planes)
    return planes


def clear_weight_planes():
    """Forget every cached bf16x3 weight plane (next to pipeline.clear_loop_cache / engine.clear_cache).  The cache is
    keyed on the weight tensor's identity and ``_version``: call this after writing a weight through ``.data`` or
    through another view of its storage, which bump neither."""
    _W3.clear()


def set_f32_mode(mode):
    """How fp32 GEMM / conv3x3 operands reach the matrix cores (include/sdamd.h, sdamd_set_f32_mode): "bf16x3" (default:
    three-term bf16 split, six MFMA products, f32-rounding-level accuracy) or "exact" (f32 MFMA, an fmaf chain over k).
    Returns the previous mode's name.  Process-wide; a captured DenoiseLoop keeps the mode it was captured under."""
    old = _lib.lib().sdamd_set_f32_mode(F32_MODES[mode])
    return "bf16x3" if old else "exact"


def get_f32_mode():
    return "bf16x3" if _lib.lib().sdamd_set_f32_mode(-1) else "exact"


def gemm_supports_act(act, dtype, N):
    return bool(_lib.lib().sdamd_gemm_supports_act(int(act), _code(dtype), int(N)))


def geglu_pack(t):
    """[2F, ...] (value rows, then gate rows; diffusion.py:128 chunk(2)) -> rows interleaved in
    16-row tiles value|gate, the layout ACT_GEGLU expects of W and bias."""
    F2 = t.shape[0]
    v, g = t[:F2 // 2], t[F2 // 2:]
    tail = t.shape[1:]
    return torch.stack([v.reshape((F2 // 32, 16) + tail), g.reshape((F2 // 32, 16) + tail)], 1).reshape(t.shape).contiguous()


def _halo_tag(h):
    return "_hbwup" if h & 0x200 else "_hbw" if h & 0x100 else "_halo" if h else ""


def conv3x3(x, w, bias, residual=None, stride=1, pad_mode=0, upsample=False, act=ACT_NONE, out_f32=False, flops_cin=None,
            want_stats=False):
    """x [B,H,W,Cin] NHWC, w [Cout, 9*Cin] ([Cout][ky][kx][Cin]) -> [B,Ho,Wo,Cout].
    ``flops_cin``: the reference conv's Cin when the executed operand is its hi|lo split (2*Cin), for the profiler's
    algorithmic-FLOP credit only.  ``want_stats``: as for ``gemm`` (the result feeds a GroupNorm)."""
    _dev(x, w, bias, residual)
    assert x.is_contiguous() and x.dim() == 4 and w.is_contiguous() and w.dtype == x.dtype
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    upsample = int(upsample)           # 0 | 1 (gathered nearest-2x) | 2 (phase-decomposed, w [4][Cout][4*Cin])
    if upsample == 2:
        assert w.dim() == 3 and w.shape[0] == 4 and w.shape[2] == 4 * Cin and stride == 1 and pad_mode == 0 and residual is None
        Cout = w.shape[1]
    else:
        assert w.shape[1] == 9 * Cin, (w.shape, Cin)
    Hin, Win = (H * 2, W * 2) if upsample else (H, W)
    if pad_mode == 0:
        Ho, Wo = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
    else:
        Ho, Wo = (Hin + 1 - 3) // stride + 1, (Win + 1 - 3) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == out.dtype and residual.is_contiguous()
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == Cout
    M = B * Ho * Wo
    ws = _gemm_workspace(x.device)
    nbytes = (x.numel() + w.numel()) * x.element_size() + out.numel() * out.element_size() * (1 if residual is None else 2)
    stats, wrote = _stats_request(want_stats and upsample != 2, M, Cout, x.device)
    w3 = weight_planes(w)
    # algorithmic FLOPs = the reference's nn.Conv2d (3x3 taps, un-split Cin), also for the 4/9-MAC phase form
    # (label suffix _halo / _hbw: the halo-staged kernels of gemm4.hip / gemm5.hip take the launch)
    _timed(lambda: f"conv3x3_{_dt(x.dtype)}_{_tile_tag(M, Cout, 9 * Cin, x.dtype, ws, f32_residual=residual is not None and out.dtype == torch.float32, conv=True)}"
                   + _halo_tag(_lib.lib().sdamd_conv3x3_halo(B, H, W, Cin, Cout, stride, pad_mode, upsample, _code(x.dtype), ws.numel())),
           2.0 * M * Cout * 9 * (Cin if flops_cin is None else flops_cin),
           lambda: _lib.check(_lib.lib().sdamd_conv3x3(_p(x), _p(w), _p(out), _p(bias), _p(residual), B, H, W, Cin, Cout, stride,
                                                       pad_mode, upsample, int(act), _code(x.dtype), int(out_f32),
                                                       _p(ws), ws.numel(), _p(stats), wrote, _p(w3), _stream())),
           lambda: f"M{M} N{Cout} K{9 * Cin} s{stride}{' up' if upsample else ''}{' +res' if residual is not None else ''}{' f32out' if out_f32 else ''}",
           nbytes)
    _stats_attach(out, stats, wrote)
    return out


def conv3x3_cin_small(x_nchw, w, bias, dtype, copies=1):
    """x [B,Cin(3|4),H,W] f32 NCHW, w [3,3,Cin,Cout] f32 -> [copies*B,H,W,Cout] dtype (sample n repeated at n, n+B, ...)."""
    _dev(x_nchw, w, bias)
    x_nchw = x_nchw.contiguous()
    assert x_nchw.dtype == torch.float32 and w.dtype == torch.float32 and bias.dtype == torch.float32
    B, Cin, H, W = x_nchw.shape
    assert w.shape[:3] == (3, 3, Cin) and w.is_contiguous() and copies >= 1
    Cout = w.shape[3]
    out = torch.empty((copies * B, H, W, Cout), dtype=dtype, device=x_nchw.device)
    _lib.check(_lib.lib().sdamd_conv3x3_cin_small(_p(x_nchw), _p(w), _p(bias), _p(out), B, H, W, Cin, Cout, int(copies),
                                                  _code(dtype), _stream()))
    return out


def conv3x3_cout_small(x, w_packed, bias, cout):
    """x [B,H,W,Cin] NHWC dtype, w_packed [9, Cin/V, Cout, V] dtype -> [B,Cout,H,W] f32 NCHW."""
    _dev(x, w_packed, bias)
    assert x.is_contiguous() and x.dim() == 4 and w_packed.dtype == x.dtype and w_packed.is_contiguous()
    B, H, W, Cin = x.shape
    assert w_packed.numel() == 9 * Cin * cout and bias.numel() == cout and bias.dtype == torch.float32
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sdamd_conv3x3_cout_small(_p(x), _p(w_packed), _p(bias), _p(out), B, H, W, Cin, cout,
                                                   _code(x.dtype), _stream()))
    return out


def attention(q, k, v, n_heads, causal=False, out=None):
    """q [B,Sq,C] and k, v [B,Skv,C] views (last dim contiguous; may be column slices of a fused
    projection) -> [B,Sq,C].  Head h = columns [h*d, (h+1)*d), d = C / n_heads."""
    _dev(q, k, v)
    B, Sq, C = q.shape
    Skv = k.shape[1]
    d = C // n_heads
    assert k.shape == (B, Skv, C) and v.shape == (B, Skv, C) and q.dtype == k.dtype == v.dtype
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty((B, Sq, C), dtype=q.dtype, device=q.device)
    else:
        assert out.shape == (B, Sq, C) and out.dtype == q.dtype and out.stride(2) == 1 and out.device == q.device
    _timed(f"attention_{_dt(q.dtype)}_d{d}", 4.0 * B * n_heads * Sq * Skv * d,
           lambda: _lib.check(_lib.lib().sdamd_attention(_p(q), _p(k), _p(v), _p(out), B, n_heads, Sq, Skv, d,
                                                         q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0),
                                                         v.stride(1), out.stride(0), out.stride(1), int(causal),
                                                         _code(q.dtype), _stream())),
           f"B{B} H{n_heads} Sq{Sq} Skv{Skv} d{d}", (2 * B * Sq * C + 2 * B * Skv * C) * q.element_size())
    return out
