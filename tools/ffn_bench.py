#!/usr/bin/env python3
"""Fused FFN (scream_ffn_x3_f32, tail_x3.hip) against the two launches it replaces (FFN-up EPI_RELU + FFN-down
EPI_RES_LN) on the forward's row counts; interleaved rounds in one process (cdna guide rule 24).

    python tools/ffn_bench.py [rows ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scream_amd import ops

dev = "cuda:0"
rows = [int(a) for a in sys.argv[1:]] or [333312, 166656, 83328]
g = torch.Generator().manual_seed(0)
W1 = (torch.randn(1024, 256, generator=g) / 16).to(dev)
W2 = (torch.randn(256, 1024, generator=g) / 32).to(dev)
gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
img, p1, p2 = ops.pack_ffn(W1, W2), ops.split_planes(W1), ops.split_planes(W2)
for M in rows:
    m1 = torch.randn(M, 256, generator=g).to(dev)
    x = torch.randn(M, 256, generator=g).to(dev)
    hid = torch.empty(M, 1024, device=dev)
    y = torch.empty(M, 256, device=dev)

    def two():
        ops.gemm_x3(m1, p1, ops.EPI_RELU, out=hid)
        ops.gemm_x3(hid, p2, ops.EPI_RES_LN, residual=x, gamma=gam, beta=bet, out=y)

    def fused():
        ops.ffn_x3(m1, img, x, gam, bet, out=y)

    res = {"two": [], "fused": []}
    for _ in range(3):
        two(); fused()
    for rnd in range(5):
        for name, fn in (("two", two), ("fused", fused)):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            b.record()
            torch.cuda.synchronize()
            res[name].append(a.elapsed_time(b) / 10)
    fl = 2.0 * M * 256 * 1024 * 2
    for name in ("two", "fused"):
        v = sorted(res[name])
        print("M=%7d %-6s median %.4f ms  min %.4f ms   %.1f TFLOP/s fp32-equivalent (median)" % (M, name, v[len(v) // 2], v[0], fl / (v[len(v) // 2] * 1e-3) / 1e12))
