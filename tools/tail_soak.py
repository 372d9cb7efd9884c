#!/usr/bin/env python3
"""Soak test of the fused layer tail (tail_split.hip: hand-counted vector-memory waits, a ring of LDS-DMA stages, operands
requested a stage ahead), a random one of its two operand splits -- and of the fp16 kernel's three forms (plain, with the next
layer's query stages, with its own query stages in front) -- per draw: random row counts and cloud partitions on two
streams at once for a fixed wall time.  Every draw is checked against the unfused chain (attention apply + merge GEMM +
FFN-up + FFN-down: same arithmetic and exponents, other summation order) to a tolerance, and run twice -- the two runs must
agree bit for bit.  usage: tail_soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scream_amd import ops, scales
dev = "cuda:0"; secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0); g = torch.Generator(device=dev).manual_seed(0)
Wqkv = torch.randn(768, 256, device=dev, generator=g) / 16; Wm = torch.randn(256, 256, device=dev, generator=g) / 16
W1 = torch.randn(1024, 256, device=dev, generator=g) / 16; W2 = torch.randn(256, 1024, device=dev, generator=g) / 32
g1, b1, g2, b2 = (torch.randn(256, device=dev, generator=g) for _ in range(4))
XMAX = 6.0  # the block inputs below are clamped to it, which bounds every operand without looking at the data
A_EXP = scales.exp_for(XMAX)
Wv = torch.cat([Wqkv[384:512], Wqkv[640:768]])  # the value rows of [q | k0-3 | v0-3 | k4-7 | v4-7]
EX = ops.tail_exps(**scales.tail_exps(Wm, W1, W2, g1, b1, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wqkv[:256].abs().sum(dim=1).max())))
P = {sp: dict(q=ops.pack_w(Wqkv, sp), m=ops.pack_w(Wm, sp), w1=ops.pack_w(W1, sp), w2=ops.pack_w(W2, sp), img=ops.pack_tail(Wm, W1, W2, sp, EX))
     for sp in (ops.SPLIT_H2, ops.SPLIT_BF3)}
# a third of the draws: the fp16 kernel with the NEXT layer's query projection behind norm2 (eight more ring stages, Q' written in place)
Wqn = torch.randn(256, 256, device=dev, generator=g) / 16
E_Y, E_WQ = scales.exp_for(scales.ln_bound(g2, b2)), scales.w_exp(Wqn)
EXQ = ops.tail_exps(e_y=E_Y, e_wq=E_WQ, **scales.tail_exps(Wm, W1, W2, g1, b1, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wqkv[:256].abs().sum(dim=1).max())))
IMG_Q, PQN = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, EXQ, Wq_next=Wqn), ops.pack_w(Wqn, ops.SPLIT_H2, E_WQ)
# a quarter of the draws: the fp16 kernel with its OWN query projection in front (tail_kernel<.., QF>: Q' never exists in memory)
EXF = ops.tail_exps(e_x=A_EXP, e_wq=scales.w_exp(Wqkv[:256]), **scales.tail_exps(Wm, W1, W2, g1, b1, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wqkv[:256].abs().sum(dim=1).max())))
IMG_F = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, EXF, Wq_own=Wqkv[:256].contiguous())
FR = ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG
KINDS = [int(k) for k in os.environ.get("T_KINDS", "0,1,2").split(",")]  # 0 fp16 plain, 1 bf16, 2 fp16 + next layer's queries, 3 fp16 + own queries (experimental kernel: NOT repeatable, see model.py q_first)
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
        tiles = np.repeat(np.arange(n_clouds), np.diff(bounds)).astype(np.int32)
        row0 = (bounds[:-1] * 128).astype(np.int32)
        lens = (np.diff(bounds) * 128 - rng.integers(0, 128, n_clouds)).astype(np.int32)
        M = n_tiles * 128
        with torch.cuda.stream(st):
            tc, cr, cl = torch.from_numpy(tiles).to(dev), torch.from_numpy(row0).to(dev), torch.from_numpy(lens).to(dev)
            x = torch.randn(M, 256, device=dev).clamp_(-XMAX, XMAX)
            xf = ops.act_layout(x, True)
            kind = int(rng.choice(KINDS))
            p = P[(ops.SPLIT_H2, ops.SPLIT_BF3, ops.SPLIT_H2, ops.SPLIT_H2)[kind]]
            img = p["img"]
            Qf, part = ops.gemm_qkv(xf, p["q"], 256, tc, cr, cl, 0, FR, a_exp=A_EXP)
            kvi = ops.kv_finalize_image(part, cr, cl, 0, 0, n_clouds, n_clouds, split=(ops.SPLIT_H2, ops.SPLIT_BF3, ops.SPLIT_H2, ops.SPLIT_H2)[kind])
            if kind == 2:
                qa, qb = Qf.clone(), Qf.clone()
                y1 = ops.layer_tail(qa, kvi, tc, 0, cl, xf, IMG_Q, g1, b1, g2, b2, q_next=qa)
                y2 = ops.layer_tail(qb, kvi, tc, 0, cl, xf, IMG_Q, g1, b1, g2, b2, q_next=qb)
                qref = ops.gemm_split(y1, PQN, ops.EPI_ELU1, n_act=256, layout=FR, a_exp=E_Y)
            elif kind == 3 and os.environ.get("T_DUMP") == "1":  # T_QF_DUMP = 4 / 5 builds: an intermediate of every tile into q_next
                IMG_F.next_q = True
                d1, d2 = torch.zeros_like(xf), torch.zeros_like(xf)
                y1 = ops.layer_tail(None, kvi, tc, 0, cl, xf, IMG_F, g1, b1, g2, b2, q_next=d1)
                y2 = ops.layer_tail(None, kvi, tc, 0, cl, xf, IMG_F, g1, b1, g2, b2, q_next=d2)
                if not torch.equal(y1, y2):
                    ry = (ops.act_layout(y1, False) != ops.act_layout(y2, False)).any(dim=1).nonzero().flatten().cpu().numpy()
                    dd = (ops.act_layout(d1, False) != ops.act_layout(d2, False))
                    rd = dd.any(dim=1).nonzero().flatten().cpu().numpy(); fd = dd.any(dim=0).nonzero().flatten().cpu().numpy()
                    print("DUMP tiles=%d: y differs in %d rows (first %d); the dumped intermediate differs in %d rows (first %s), %d features, chunks %s, max %.3g"
                          % (M // 128, len(ry), ry[0], len(rd), rd[:1], len(fd), np.unique(fd // 32), float((d1 - d2).abs().max())), flush=True)
            elif kind == 3:
                y1 = ops.layer_tail(None, kvi, tc, 0, cl, xf, IMG_F, g1, b1, g2, b2)
                y2 = ops.layer_tail(None, kvi, tc, 0, cl, xf, IMG_F, g1, b1, g2, b2)
            else:
                y1 = ops.layer_tail(Qf, kvi, tc, 0, cl, xf, img, g1, b1, g2, b2)
                y2 = ops.layer_tail(Qf, kvi, tc, 0, cl, xf, img, g1, b1, g2, b2)
            kv = ops.kv_finalize(part, cr, cl, 0, 0, n_clouds, n_clouds)
            att = ops.attn_apply(ops.act_layout(Qf, False), 256, kv, tc, 0, cl, M)
            m1 = ops.gemm_split(att, p["m"], ops.EPI_RES_LN, residual=x, gamma=g1, beta=b1, a_exp=EX.e_att)
            hid = ops.gemm_split(m1, p["w1"], ops.EPI_RELU, a_exp=EX.e_m1)
            ref = ops.gemm_split(hid, p["w2"], ops.EPI_RES_LN, residual=x, gamma=g2, beta=b2, a_exp=EX.e_h)
            yr = ops.act_layout(y1, False)
            valid = torch.zeros(M, dtype=torch.bool, device=dev)
            for r0, ln in zip(row0.tolist(), lens.tolist()):
                valid[r0:r0 + ln] = True
            err = ((yr - ref).abs() * valid[:, None]).max()
            same = torch.equal(y1, y2)
            if kind == 2:  # the projected queries: against the projection GEMM of the same y, and repeatable
                err = torch.maximum(err, ((ops.act_layout(qa, False) - ops.act_layout(qref, False)).abs() * valid[:, None]).max() * 100)  # (tolerance 1e-5)
                same = same and torch.equal(qa, qb)
            if os.environ.get("T_DIAG") == "1":  # where two runs differ (costly: a synchronisation per draw)
                da = (ops.act_layout(y1, False) != ops.act_layout(y2, False))
                if bool(da.any()):
                    rows = da.any(dim=1).nonzero().flatten().cpu().numpy(); feats = da.any(dim=0).nonzero().flatten().cpu().numpy()
                    tl = np.unique(rows // 128)
                    y3 = ops.layer_tail(None, kvi, tc, 0, cl, xf, IMG_F, g1, b1, g2, b2) if kind == 3 else y1
                    print("DIFF kind=%d tiles=%d clouds=%d: %d rows; tiles %s (mod 256 %s, round %s); waves %s; rows-in-group %d..%d; feats %d (%s) chunks %s; max %.3g; y3==y1 %s y3==y2 %s"
                          % (kind, M // 128, n_clouds, len(rows), tl[:10], (tl % 256)[:10], (tl // 256)[:10], np.unique((rows % 128) // 32), (rows % 32).min(), (rows % 32).max(), len(feats), feats[:12], np.unique(feats // 32),
                             float((y1 - y2).abs().max()), bool(torch.equal(y3, y1)), bool(torch.equal(y3, y2))), flush=True)
            jobs.append((M, n_clouds, err, same, kind))
    torch.cuda.synchronize()
    if n % 100 == 0: print("progress", n, "%.0f s" % (time.time() - t0), flush=True)
    for M, nc, err, same, kind in jobs:
        e = float(err); worst = max(worst, e)
        if not same or not (e < 1e-3):
            print("MISMATCH kind=%d M=%d clouds=%d max|fused - unfused|=%g bitwise-repeatable=%s after %d draws; jobs %s" % (kind, M, nc, e, same, n, [(j[0], j[1], j[4]) for j in jobs]), flush=True)
            if os.environ.get("T_DIAG") != "1": sys.exit(1)
        n += 1
print("tail soak ok: %d draws in %.0f s on two streams, worst |fused - unfused| = %.3g, every draw bitwise repeatable" % (n, time.time() - t0, worst))
