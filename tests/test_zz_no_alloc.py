"""sdamd_groupnorm_nhwc_noalloc: the GroupNorm entry point for callers that cannot allocate (include/sdamd.h).

Called straight through the C ABI (ctypes), checked bit for bit against sdamd_groupnorm_nhwc (the existing entry
point with a caller workspace; same kernels) and against torch's CPU group_norm, on calls in a row -- the second with
a smaller input, so stale partials of the first call would show.  Collected last (file name): a failure here must
not hide the other GPU tests behind ``pytest -x``.
"""
import pytest
import torch

import sd_amd  # noqa: F401
from sd_amd import _lib
from conftest import hin, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _call(fn_noalloc, x, gamma, beta, eps, silu, out_dtype):
    """(no-alloc result, workspace result); x [B,HW,C] on the device."""
    L = _lib.lib()
    B, HW, C = x.shape
    code = {torch.float32: 0, torch.bfloat16: 1}
    y_a = torch.empty((B, HW, C), dtype=out_dtype, device=x.device)
    y_b = torch.empty_like(y_a)
    ws = torch.empty(L.sdamd_groupnorm_ws_bytes(B), dtype=torch.uint8, device=x.device)
    st = torch.cuda.current_stream().cuda_stream
    args = (B, HW, C, float(eps), int(silu), code[x.dtype], code[out_dtype], st)
    if fn_noalloc:
        _lib.check(L.sdamd_groupnorm_nhwc_noalloc(x.data_ptr(), y_a.data_ptr(), gamma.data_ptr(), beta.data_ptr(), *args))
    _lib.check(L.sdamd_groupnorm_nhwc(x.data_ptr(), y_b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), *args))
    torch.cuda.synchronize()
    return y_a, y_b


def test_noalloc_groupnorm_two_calls_second_smaller():
    L = _lib.lib()
    assert L.sdamd_noalloc_init() == 0
    cases = [   # (B, HW, C, in dtype, out dtype, silu): sizes that take the stats + apply path (the scratch user)
        (4, 64 * 64, 320, torch.float32, torch.bfloat16, 1),
        (2, 32 * 32, 640, torch.float32, torch.float32, 0),      # second call: smaller batch, fewer blocks per image
        (3, 64 * 64, 128, torch.bfloat16, torch.bfloat16, 1),
        (1, 40 * 40, 320, torch.float32, torch.float32, 1),
    ]
    for i, (B, HW, C, dt, odt, silu) in enumerate(cases):
        x = (hin(f"noalloc.x.{i}", (B, HW, C), 2.0) + 0.25).to(dt).to(DEV).contiguous()
        gamma = (1.0 + 0.1 * hin(f"noalloc.g.{i}", (C,))).to(DEV)
        beta = (0.1 * hin(f"noalloc.b.{i}", (C,))).to(DEV)
        got, want = _call(True, x, gamma, beta, 1e-5, silu, odt)
        assert torch.equal(got, want), f"case {i}: no-alloc entry point differs from the workspace entry point"
        # and both against the oracle (NCHW reference GroupNorm)
        xo = x.float().cpu().permute(0, 2, 1).reshape(B, C, HW, 1)
        ref = torch.nn.functional.group_norm(xo, 32, gamma.cpu(), beta.cpu(), 1e-5)
        if silu:
            ref = torch.nn.functional.silu(ref)
        ref = ref.reshape(B, C, HW).permute(0, 2, 1)
        tol = 1e-5 if odt == torch.float32 and dt == torch.float32 else 1e-2
        assert rel_l2(got.float().cpu(), ref) < tol, (i, rel_l2(got.float().cpu(), ref))


def test_noalloc_groupnorm_rejects_oversize_batch():
    L = _lib.lib()
    maxb = L.sdamd_groupnorm_noalloc_max_batch()
    B, HW, C = maxb + 1, 16, 32
    x = torch.zeros((B, HW, C), device=DEV)
    y = torch.empty_like(x)
    g = torch.ones(C, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.sdamd_groupnorm_nhwc_noalloc(x.data_ptr(), y.data_ptr(), g.data_ptr(), g.data_ptr(), B, HW, C, 1e-5, 0, 0, 0, st)
    assert rc != 0 and b"exceeds the built-in scratch" in L.sdamd_last_error()


def test_noalloc_groupnorm_under_graph_capture():
    """The callers it is for: captured into a hipGraph with no workspace argument, replayed twice."""
    L = _lib.lib()
    assert L.sdamd_noalloc_init() == 0
    B, HW, C = 2, 64 * 64, 320
    x = hin("noalloc.cap.x", (B, HW, C), 2.0).to(DEV).contiguous()
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    y = torch.empty_like(x)
    _, want = _call(False, x, gamma, beta, 1e-5, 1, torch.float32)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(L.sdamd_groupnorm_nhwc_noalloc(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B, HW, C,
                                                  1e-5, 1, 0, 0, st))
    for _ in range(2):
        y.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)
