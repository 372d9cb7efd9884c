#!/usr/bin/env python3
"""Soak test of the ring projection kernel (csrc/proj_ring.hip: LDS-DMA ring with counted vector-memory waits, epilogues riding
under the next stage, contiguous unit ranges per block): random row counts, cloud partitions and row bases, the q | k | v form and
the key/value-only form of 1-6 layers, on two streams at once for a fixed wall time.  Every draw runs twice -- the two runs must
agree bit for bit -- and is checked against the 8-wave GEMM on the same operands (Q' bit for bit, the K^T V partials summed per
cloud to fp32 rounding).  usage: proj_soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scream_amd import ops, scales
dev = "cuda:0"; secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0); g = torch.Generator(device=dev).manual_seed(0)
XMAX = 6.0
A_EXP = scales.exp_for(XMAX)
FORMS = {}
for L in (0, 1, 2, 6):  # 0: q | k | v of one layer; L > 0: key/value-only, L layers stacked
    N, n_q = (768, 256) if L == 0 else (512 * L, 0)
    W = torch.randn(N, 256, device=dev, generator=g) / 16
    w_exp = scales.w_exp(W)
    rl = W.abs().sum(dim=1).cpu()[n_q:].view(-1, 2, 128)
    FORMS[L] = dict(N=N, n_q=n_q, P=ops.pack_proj(W, n_q, ops.SPLIT_H2, w_exp), G=ops.pack_w(W, ops.SPLIT_H2, w_exp),
                    k_exp=scales.exp_for(1.0 + XMAX * float(rl[:, 0].max())), v_exp=scales.exp_for(XMAX * float(rl[:, 1].max())))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
t0 = time.time(); n = 0; worst = 0.0
while time.time() - t0 < secs:
    jobs = []
    for st in streams:
        n_tiles = int(rng.integers(1, 1400))
        n_clouds = int(rng.integers(1, min(n_tiles, 64) + 1))
        cuts = np.sort(rng.choice(np.arange(1, n_tiles), n_clouds - 1, replace=False)) if n_clouds > 1 else np.array([], dtype=int)
        bounds = np.concatenate([[0], cuts, [n_tiles]])
        base_tiles = int(rng.integers(0, 5))  # rows in front of the launch's first row (another cloud's): row_base = 128 base_tiles
        tiles = np.concatenate([np.full(base_tiles, n_clouds), np.repeat(np.arange(n_clouds), np.diff(bounds))]).astype(np.int32)
        row0 = np.concatenate([(bounds[:-1] + base_tiles) * 128, [0]]).astype(np.int32)
        lens = np.concatenate([np.diff(bounds) * 128 - rng.integers(0, 128, n_clouds), [base_tiles * 128]]).astype(np.int32)
        M, base = n_tiles * 128, base_tiles * 128
        F = FORMS[int(rng.choice([0, 0, 1, 2, 6]))]
        if F["N"] > 1024 and n_tiles > 500:
            F = FORMS[1]
        with torch.cuda.stream(st):
            tc, cr, cl = torch.from_numpy(tiles).to(dev), torch.from_numpy(row0).to(dev), torch.from_numpy(lens).to(dev)
            xf = ops.act_layout(torch.randn(M, 256, device=dev).clamp_(-XMAX, XMAX), True)
            kw = dict(a_exp=A_EXP, k_exp=F["k_exp"], v_exp=F["v_exp"])
            q1, p1 = ops.proj_qkv(xf, F["P"], tc, cr, cl, base, **kw)
            q2, p2 = ops.proj_qkv(xf, F["P"], tc, cr, cl, base, **kw)
            qg, pg = ops.gemm_qkv(xf, F["G"], F["n_q"], tc, cr, cl, base, layout=ops.LAYOUT_A_FRAG | (ops.LAYOUT_C_FRAG if F["n_q"] else 0), **kw)
            layers = [(p1, pg)] if p1.dim() == 3 else [(p1[l], pg[l]) for l in range(p1.shape[0])]
            kvs = [(ops.kv_finalize(a, cr, cl, base, 0, n_clouds, n_clouds + 1), ops.kv_finalize(b, cr, cl, base, 0, n_clouds, n_clouds + 1)) for a, b in layers]
        jobs.append((q1, q2, qg, p1, p2, kvs, M, F["N"]))
    torch.cuda.synchronize()
    for q1, q2, qg, p1, p2, kvs, M, N in jobs:
        assert torch.equal(p1, p2) and (q1 is None or torch.equal(q1, q2)), "ring projection not repeatable (M=%d, N=%d)" % (M, N)
        assert q1 is None or torch.equal(q1, qg), "Q' differs from the 8-wave GEMM's (M=%d)" % M
        for a, b in kvs:
            err = float(((a - b).abs() / (b.abs() + 1e-2)).max())
            worst = max(worst, err)
            assert err < 2e-4, "K^T V differs from the 8-wave GEMM's: %g (M=%d, N=%d)" % (err, M, N)
        n += 1
print("proj_soak: %d draws in %.0f s on two streams, all bitwise repeatable, Q' identical to the 8-wave GEMM's, worst K^T V rel diff %.3g" % (n, time.time() - t0, worst))
