#!/usr/bin/env python3
"""Time scream_gemm_f32 alone on the shapes of one forward pass (HIP events, interleaved rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 327680
dev = "cuda:0"
shapes = [("qkv  N=768 K=256 elu", 768, 256, ops.EPI_ELU1), ("ffn1 N=1024 K=256 relu", 1024, 256, ops.EPI_RELU),
          ("ffn2 N=256 K=1024 res+ln", 256, 1024, ops.EPI_RES_LN), ("merge N=256 K=256 res+ln", 256, 256, ops.EPI_RES_LN),
          ("plain N=256 K=256", 256, 256, ops.EPI_NONE)]
g = torch.Generator(device=dev).manual_seed(0)
bufs = {}
for name, N, K, epi in shapes:
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    out = torch.empty(M, N, device=dev)
    res = torch.randn(M, 256, device=dev, generator=g) if epi == ops.EPI_RES_LN else None
    gam = torch.ones(256, device=dev) if epi == ops.EPI_RES_LN else None
    bufs[name] = (A, W, out, res, gam)

def run(name, N, K, epi):
    A, W, out, res, gam = bufs[name]
    ops.gemm_f32(A, W, epi, n_act=512 if epi == ops.EPI_ELU1 else 0, residual=res, gamma=gam, beta=gam, out=out)

import ctypes
from scream_amd import _lib
lib = _lib.load()
for mode in [0]:
  for s in shapes:
    run(*s)
  torch.cuda.synchronize()
  times = {s[0]: [] for s in shapes}
  bench_mode = True
  for rnd in range(5):
    for s in shapes:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(*s)
        e1.record()
        torch.cuda.synchronize()
        times[s[0]].append(e0.elapsed_time(e1) / 5)
  for name, N, K, epi in shapes:
    t = sorted(times[name])
    med = t[len(t) // 2]
    print("%-28s M=%d  median %.3f ms  min %.3f ms  %.1f TFLOP/s (median)" % (name, M, med, t[0], 2.0 * M * N * K / med / 1e9))
