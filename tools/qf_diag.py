#!/usr/bin/env python3
"""Diagnostic: run the layer tail with its own query projection (QF) repeatedly on one large batch and report WHERE two runs differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scream_amd import ops, scales
dev = "cuda:0"; g = torch.Generator(device=dev).manual_seed(0)
M = int(os.environ.get("T_M", 175232)); XMAX = 6.0
Wqkv = torch.randn(768, 256, device=dev, generator=g) / 16; Wm = torch.randn(256, 256, device=dev, generator=g) / 16
W1 = torch.randn(1024, 256, device=dev, generator=g) / 16; W2 = torch.randn(256, 1024, device=dev, generator=g) / 32
g1, b1, g2, b2 = (torch.randn(256, device=dev, generator=g) for _ in range(4))
A_EXP = scales.exp_for(XMAX); Wq = Wqkv[:256].contiguous(); Wv = torch.cat([Wqkv[384:512], Wqkv[640:768]])
exd = scales.tail_exps(Wm, W1, W2, g1, b1, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wq.abs().sum(dim=1).max()))
own = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, ops.tail_exps(e_x=A_EXP, e_wq=scales.w_exp(Wq), **exd), Wq_own=Wq)
rng = np.random.default_rng(int(os.environ.get("T_SEED", 11)))
bad = 0
for it in range(int(os.environ.get("T_ITERS", 400))):
    n_tiles = int(rng.integers(1, 1400))
    n_clouds = int(rng.integers(1, min(n_tiles, 64) + 1))
    cuts = np.sort(rng.choice(np.arange(1, n_tiles), n_clouds - 1, replace=False)) if n_clouds > 1 else np.array([], dtype=int)
    bounds = np.concatenate([[0], cuts, [n_tiles]])
    tiles = np.repeat(np.arange(n_clouds), np.diff(bounds)).astype(np.int32)
    row0_ = (bounds[:-1] * 128).astype(np.int32)
    lens_ = (np.diff(bounds) * 128 - rng.integers(0, 128, n_clouds)).astype(np.int32)
    M = n_tiles * 128
    tc, row0, lens = torch.from_numpy(tiles).to(dev), torch.from_numpy(row0_).to(dev), torch.from_numpy(lens_).to(dev)
    x = torch.randn(M, 256, device=dev).clamp_(-XMAX, XMAX); xf = ops.act_layout(x, True)
    Qf, part = ops.gemm_qkv(xf, ops.pack_w(Wqkv, ops.SPLIT_H2), 256, tc, row0, lens, 0, 3, a_exp=A_EXP)
    kvi = ops.kv_finalize_image(part, row0, lens, 0, 0, n_clouds, n_clouds, split=ops.SPLIT_H2)
    ya = ops.layer_tail(None, kvi, tc, 0, lens, xf, own, g1, b1, g2, b2)
    yb = ops.layer_tail(None, kvi, tc, 0, lens, xf, own, g1, b1, g2, b2)
    yc = ops.layer_tail(None, kvi, tc, 0, lens, xf, own, g1, b1, g2, b2)
    ref, y, y3 = (ops.act_layout(t, False) for t in (ya, yb, yc))
    d = (y != ref)
    if bool(d.any()):
        rows = d.any(dim=1).nonzero().flatten().cpu().numpy()
        feats = d.any(dim=0).nonzero().flatten().cpu().numpy()
        tl = np.unique(rows // 128)
        which = "first run is the odd one" if bool((y == y3).all()) else "second run is the odd one" if bool((ref == y3).all()) else "all three differ"
        print("draw %d (tiles %d, clouds %d): %d rows differ; tiles %s (tile %% 256 = %s, round %s); clouds %s; waves %s; rows in group %s..%s; features: %d of 256 (%s...); max diff %.3g; %s"
              % (it, n_tiles, n_clouds, len(rows), tl[:8], (tl % 256)[:8], (tl // 256)[:8], np.unique(tiles[tl])[:8], np.unique((rows % 128) // 32), (rows % 32).min(), (rows % 32).max(),
                 len(feats), feats[:8], float((y - ref).abs().max()), which), flush=True)
        print("     lens of those clouds:", lens_[np.unique(tiles[tl])][:8], "cloud tile ranges:", [(int(bounds[c]), int(bounds[c + 1])) for c in np.unique(tiles[tl])[:4]], flush=True)
        bad += 1
        if bad >= 6: break
print("done: %d differing draws" % bad)
