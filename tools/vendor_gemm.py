#!/usr/bin/env python3
"""Context number: the vendor fp32 GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the forward's GEMM shapes, no epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import ops
dev = "cuda:0"; M = 327680
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K in [("qkv", 768, 256), ("ffn1", 1024, 256), ("ffn2", 256, 1024), ("merge", 256, 256)]:
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    Wt = W.t().contiguous(); Wp = ops.pack_w(W); o = torch.empty(M, N, device=dev)
    def t(fn):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
        return sorted(ts)[2]
    tv = t(lambda: torch.matmul(A, Wt, out=o)); t3 = t(lambda: ops.gemm_split(A, Wp, out=o)); tf = t(lambda: ops.gemm_f32(A, W, out=o))
    f = 2.0 * M * N * K / 1e9
    print("%-6s N=%4d K=%4d  vendor fp32 %.3f ms %.1f TF | gemm_x3 %.3f ms %.1f TF | gemm_f32 %.3f ms %.1f TF" % (name, N, K, tv, f / tv, t3, f / t3, tf, f / tf))

print("vendor bf16 GEMM (bf16 in, bf16 out; one of the six products of the split):")
for name, N, K in [("qkv", 768, 256), ("ffn1", 1024, 256), ("ffn2", 256, 1024), ("merge", 256, 256)]:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); Wt = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16().t().contiguous()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: torch.matmul(A, Wt, out=o)
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    tb = sorted(ts)[2]; f = 2.0 * M * N * K / 1e9
    print("%-6s N=%4d K=%4d  %.3f ms %.0f TF  -> six of them %.3f ms = %.1f TF fp32-equivalent" % (name, N, K, tb, f / tb, 6 * tb, f / (6 * tb)))
