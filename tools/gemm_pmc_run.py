#!/usr/bin/env python3
"""Launch each forward GEMM shape a few times on the shipped x3 kernel (target of rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import ops
dev = "cuda:0"; M = 327680
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, epi in [("qkv", 768, 256, ops.EPI_ELU1), ("ffn1", 1024, 256, ops.EPI_RELU), ("ffn2", 256, 1024, ops.EPI_RES_LN), ("merge", 256, 256, ops.EPI_RES_LN)]:
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    Wp = ops.pack_w(W); o = torch.empty(M, N, device=dev); r = torch.randn(M, 256, device=dev, generator=g); gam = torch.ones(256, device=dev)
    for _ in range(4):
        ops.gemm_split(A, Wp, epi, n_act=512 if epi == ops.EPI_ELU1 else 0, residual=r, gamma=gam, beta=gam, out=o)
    torch.cuda.synchronize()
