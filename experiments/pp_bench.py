"""Ping-pong GEMM / conv kernels (gemm3.hip) against the ring kernels (gemm2.hip): bit-equality and time, same process,
interleaved rounds, operands rotated over several buffer sets (cold-ish: the in-situ state is 'written by the previous
kernel').  usage: pp_bench.py [pp_bits=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_amd
from sd_amd import ops, _lib
torch.set_grad_enabled(False)
dev = "cuda"
PP = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = _lib.lib()

def setpp(bits):
    L.sdamd_set_gemm_impl(bits << 20)

def rnd(*shape, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)

def run_case(name, make, call, flops, nset=3, rounds=5, iters=8):
    sets = [make() for _ in range(nset)]
    outs = {}
    for bits in (0, PP):
        setpp(bits)
        outs[bits] = call(*sets[0]).clone()
    eq = torch.equal(outs[0], outs[PP])
    md = (outs[0].float() - outs[PP].float()).abs().max().item()
    times = {0: [], PP: []}
    for r in range(rounds):
        for bits in (0, PP):
            setpp(bits)
            call(*sets[0]); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters):
                call(*sets[i % nset])
            e.record(); torch.cuda.synchronize()
            times[bits].append(s.elapsed_time(e) / iters * 1e3)
    setpp(0)
    a, b = sorted(times[0])[rounds // 2], sorted(times[PP])[rounds // 2]
    print(f"{name:46s} ring {a:8.1f} us {flops / a / 1e6:7.1f} TF | pp {b:8.1f} us {flops / b / 1e6:7.1f} TF | {100 * (b / a - 1):+6.1f} %  bit-equal {eq} maxdiff {md:.2e}", flush=True)

def gemm_case(M, N, K, geglu=False, res=False, out_f32=False):
    def make():
        a = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if res else None
        return a, w, b, r
    def call(a, w, b, r):
        return ops.gemm(a, w, b, r, ops.ACT_GEGLU if geglu else ops.ACT_NONE, out_f32=out_f32 or res)
    run_case(f"gemm M{M} N{N} K{K}{' geglu' if geglu else ''}{' +res' if res else ''}{' f32out' if out_f32 else ''}", make, call, 2.0 * M * N * K)

def conv_case(B, H, Cin, Cout, res=False):
    def make():
        x = rnd(B, H, H, Cin); w = rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5); b = torch.randn(Cout, device=dev)
        r = torch.randn(B, H, H, Cout, device=dev) if res else None
        return x, w, b, r
    def call(x, w, b, r):
        return ops.conv3x3(x, w, b, r, out_f32=res)
    run_case(f"conv B{B} {H}x{H} Cin{Cin} Cout{Cout}{' +res' if res else ''}", make, call, 2.0 * B * H * H * Cout * 9 * Cin)

if PP & 1:
    gemm_case(65536, 2560, 320, geglu=True)
    gemm_case(16384, 5120, 640, geglu=True)
    gemm_case(4096, 10240, 1280, geglu=True)
    gemm_case(8192, 2560, 320, geglu=True)
    gemm_case(16384, 3072, 768)          # CLIP-ish / 256-wide
    conv_case(8, 64, 512, 512)           # VAE 64x64 x512
    conv_case(8, 128, 512, 512, res=True)
    conv_case(4, 256, 256, 256)
    conv_case(2, 512, 128, 128)
if PP & 2:
    conv_case(16, 64, 320, 320)
    conv_case(16, 64, 320, 320, res=True)
    conv_case(16, 64, 640, 320)
    conv_case(16, 64, 960, 320)
    conv_case(16, 32, 640, 640)
    conv_case(16, 32, 1280, 640)
    gemm_case(65536, 320, 1280, res=True)
    gemm_case(65536, 960, 320)
    gemm_case(65536, 320, 320, res=True)
