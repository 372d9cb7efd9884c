#!/usr/bin/env python3
"""Experiment: 3xbf16-split GEMM vs the fp32-MFMA GEMM (accuracy against fp64 and speed)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import _lib, ops
dev = "cuda:0"
planes = ops.pack_w

def x3(A, Wp, N, epi=ops.EPI_NONE, **kw):
    return ops.gemm_split(A, Wp, epi, **kw)

g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(512, 256, 256), (512, 1024, 1024)]:
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    ref = A.double() @ W.double().t()
    scale = A.double().abs() @ W.double().abs().t()
    e3 = ((x3(A, planes(W), N).double() - ref).abs() / scale).max().item()
    e32 = ((ops.gemm_f32(A, W).double() - ref).abs() / scale).max().item()
    print("accuracy M%d N%d K%d: max err/scale  x3 %.3e   fp32-mfma %.3e" % (M, N, K, e3, e32))
M = 327680
for name, N, K, epi in [("qkv", 768, 256, ops.EPI_ELU1), ("ffn1", 1024, 256, ops.EPI_RELU), ("ffn2", 256, 1024, ops.EPI_RES_LN), ("merge", 256, 256, ops.EPI_RES_LN)]:
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    Wp = planes(W)
    res = torch.randn(M, 256, device=dev, generator=g) if epi == ops.EPI_RES_LN else None
    gam = torch.ones(256, device=dev) if epi == ops.EPI_RES_LN else None
    o = torch.empty(M, N, device=dev)
    kw = dict(n_act=512 if epi == ops.EPI_ELU1 else 0, residual=res, gamma=gam, beta=gam, out=o)
    for fn, tag in ((lambda: x3(A, Wp, N, epi, **kw), "x3  "), (lambda: ops.gemm_f32(A, W, epi, **kw), "fp32")):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
        t = sorted(ts)[2]
        print("%-6s %s N=%4d K=%4d  %.3f ms  %.1f TFLOP/s (fp32-equivalent)" % (name, tag, N, K, t, 2.0 * M * N * K / t / 1e9))
