#!/usr/bin/env python3
"""Soak test: the split GEMM (a random one of its two operand splits per draw: 2 x fp16 / 3 x bf16) against the fp32-MFMA GEMM
on random (M, epilogue) draws for a fixed wall time, on two streams at once (the lanes configuration).  A mismatch is re-checked against torch on the default stream to tell
which kernel is off.  usage: gemm_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scream_amd import ops, scales
dev = "cuda:0"; secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0); g = torch.Generator(device=dev).manual_seed(0)
Ws = {(N, K): torch.randn(N, K, device=dev, generator=g) / K ** 0.5 for N, K in [(256, 256), (1024, 256), (256, 1024), (768, 256), (512, 64)]}
Wp = {(k, sp): ops.pack_w(w, sp) for k, w in Ws.items() for sp in (ops.SPLIT_H2, ops.SPLIT_BF3)}
A_EXP = scales.exp_for(8.0)  # the activations below are clamped to |a| <= 8: the fp16 split's exponent without a device sync
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
t0 = time.time(); n = 0; worst = 0.0
while time.time() - t0 < secs:
    jobs = []
    for st in streams:
        N, K = list(Ws)[rng.integers(len(Ws))]
        M = int(rng.integers(1, 1200)) * 128
        epi = [ops.EPI_NONE, ops.EPI_RELU, ops.EPI_RES_LN][rng.integers(3)] if N == 256 else [ops.EPI_NONE, ops.EPI_RELU][rng.integers(2)]
        with torch.cuda.stream(st):
            A = torch.randn(M, K, device=dev).clamp_(-8.0, 8.0)
            sp = (ops.SPLIT_H2, ops.SPLIT_BF3)[rng.integers(2)]
            kw = {}
            if epi == ops.EPI_RES_LN:
                kw = dict(residual=torch.randn(M, 256, device=dev), gamma=torch.randn(256, device=dev), beta=torch.randn(256, device=dev))
            a = ops.gemm_split(A, Wp[((N, K), sp)], epi, a_exp=A_EXP, **kw)
            b = ops.gemm_f32(A, Ws[(N, K)], epi, **kw)
            jobs.append((M, N, K, epi, (a - b).abs().max(), b.abs().max(), A, a, b, kw))
    torch.cuda.synchronize()
    if n % 400 == 0: print("progress", n, "%.0f s" % (time.time() - t0), flush=True)
    for M, N, K, epi, err, scale, A, a, b, kw in jobs:
        e = float(err); worst = max(worst, e)
        if not (e < 2e-4 * max(1.0, float(scale))):
            ref = A @ Ws[(N, K)].t()
            if epi == ops.EPI_RELU: ref = ref.clamp_min(0)
            if epi == ops.EPI_RES_LN: ref = torch.nn.functional.layer_norm(ref + kw["residual"], (256,), kw["gamma"], kw["beta"], 1e-5)
            ea, eb = (a - ref).abs(), (b - ref).abs()
            def where(x):
                blk = x.reshape(M // 32, 32, N // 256, 256).amax(dim=(1, 3)); idx = (blk > 1e-2).nonzero()
                return "count %d first %s" % (idx.shape[0], idx[:6].tolist())
            print("MISMATCH M=%d N=%d K=%d epi=%d |x3-f32|=%g after %d pairs; jobs %s\n   x3 vs torch: max %g  bad blocks %s\n   f32 vs torch: max %g  bad blocks %s" % (
                M, N, K, epi, e, n, [(j[0], j[1], j[2], j[3]) for j in jobs], ea.max().item(), where(ea), eb.max().item(), where(eb)), flush=True)
            sys.exit(1)
        n += 1
print("soak ok: %d GEMM pairs in %.0f s, worst |x3 - f32| = %.3g" % (n, time.time() - t0, worst))
