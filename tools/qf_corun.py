#!/usr/bin/env python3
"""What has to run beside the experimental QF tail (own query projection) for two launches on the same inputs to differ?  Stream A: large QF
draws (two launches, compared bit for bit); stream B, all the time: nothing | big device-to-device copies (memory traffic, little power) |
fp16 matmuls on L2-sized operands (matrix-core power, little memory traffic) | the unfused reference chain of tools/tail_soak.py.
usage: qf_corun.py <none|copy|mm|chain> [seconds] [seed] [qf|plain|nq|x3|proj]   (the last: which kernel is drawn; default qf)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scream_amd import ops, scales
mode = sys.argv[1]; secs = float(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
kind = sys.argv[4] if len(sys.argv) > 4 else "qf"
dev = "cuda:0"; g = torch.Generator(device=dev).manual_seed(0)
Wqkv = torch.randn(768, 256, device=dev, generator=g) / 16; Wm = torch.randn(256, 256, device=dev, generator=g) / 16
W1 = torch.randn(1024, 256, device=dev, generator=g) / 16; W2 = torch.randn(256, 1024, device=dev, generator=g) / 32
g1, b1, g2, b2 = (torch.randn(256, device=dev, generator=g) for _ in range(4))
XMAX = 6.0; A_EXP = scales.exp_for(XMAX)
Wv = torch.cat([Wqkv[384:512], Wqkv[640:768]])
base = scales.tail_exps(Wm, W1, W2, g1, b1, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wqkv[:256].abs().sum(dim=1).max()))
EX = ops.tail_exps(**base); EXF = ops.tail_exps(e_x=A_EXP, e_wq=scales.w_exp(Wqkv[:256]), **base)
PQ = ops.pack_w(Wqkv, ops.SPLIT_H2); PM, P1, P2 = ops.pack_w(Wm, ops.SPLIT_H2), ops.pack_w(W1, ops.SPLIT_H2), ops.pack_w(W2, ops.SPLIT_H2)
IMG_F = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, EXF, Wq_own=Wqkv[:256].contiguous())
IMG = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, EX); IMG3 = ops.pack_tail(Wm, W1, W2, ops.SPLIT_BF3, EX); PQ3 = ops.pack_w(Wqkv, ops.SPLIT_BF3)
Wqn = torch.randn(256, 256, device=dev, generator=g) / 16
EXQ = ops.tail_exps(e_y=scales.exp_for(scales.ln_bound(g2, b2)), e_wq=scales.w_exp(Wqn), **base)
IMG_Q = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, EXQ, Wq_next=Wqn)
PP = ops.pack_proj(Wqkv, 256, ops.SPLIT_H2, scales.w_exp(Wqkv)) if kind == "proj" else None
_rl = Wqkv[256:].abs().sum(dim=1).view(-1, 2, 128)
EXK = (scales.exp_for(1.0 + XMAX * float(_rl[:, 0].max())), scales.exp_for(XMAX * float(_rl[:, 1].max())))
FR = ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
big_a, big_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)          # 256 MB each
ma, mb = torch.randn(2048, 2048, device=dev, dtype=torch.float16), torch.randn(2048, 2048, device=dev, dtype=torch.float16)
xc = torch.randn(150016, 256, device=dev).clamp_(-XMAX, XMAX)
torch.cuda.synchronize()


def corun():
    with torch.cuda.stream(sb):
        if mode == "copy":
            for _ in range(6): big_b.copy_(big_a)
        elif mode == "mm":
            for _ in range(40): torch.matmul(ma, mb)
        elif mode == "chain":
            m1 = ops.gemm_split(xc, PM, ops.EPI_RES_LN, residual=xc, gamma=g1, beta=b1, a_exp=A_EXP)
            hid = ops.gemm_split(m1, P1, ops.EPI_RELU, a_exp=EX.e_m1)
            ops.gemm_split(hid, P2, ops.EPI_RES_LN, residual=xc, gamma=g2, beta=b2, a_exp=EX.e_h)


t0 = time.time(); n = bad = 0; shown = 0
while time.time() - t0 < secs:
    n_tiles = int(rng.integers(940, 1400)); n_clouds = int(rng.integers(1, 65))
    cuts = np.sort(rng.choice(np.arange(1, n_tiles), n_clouds - 1, replace=False)) if n_clouds > 1 else np.array([], dtype=int)
    bounds = np.concatenate([[0], cuts, [n_tiles]]); tiles = np.repeat(np.arange(n_clouds), np.diff(bounds)).astype(np.int32)
    row0 = (bounds[:-1] * 128).astype(np.int32); lens = (np.diff(bounds) * 128 - rng.integers(0, 128, n_clouds)).astype(np.int32); M = n_tiles * 128
    corun()
    with torch.cuda.stream(sa):
        tc, cr, cl = torch.from_numpy(tiles).to(dev), torch.from_numpy(row0).to(dev), torch.from_numpy(lens).to(dev)
        x = torch.randn(M, 256, device=dev).clamp_(-XMAX, XMAX); xf = ops.act_layout(x, True)
        sp = ops.SPLIT_BF3 if kind == "x3" else ops.SPLIT_H2
        Qf, part = ops.gemm_qkv(xf, PQ3 if kind == "x3" else PQ, 256, tc, cr, cl, 0, FR, a_exp=A_EXP)
        kvi = ops.kv_finalize_image(part, cr, cl, 0, 0, n_clouds, n_clouds, split=sp)
        def run():
            if kind == "qf" and os.environ.get("QF_DUMP") == "1":  # T_QF_DUMP = 4 / 5 builds: an intermediate of every tile into q_next
                IMG_F.next_q = True
                d = torch.zeros_like(xf)
                return (ops.layer_tail(None, kvi, tc, 0, cl, xf, IMG_F, g1, b1, g2, b2, q_next=d), d)
            if kind == "qf": return (ops.layer_tail(None, kvi, tc, 0, cl, xf, IMG_F, g1, b1, g2, b2),)
            if kind == "plain": return (ops.layer_tail(Qf, kvi, tc, 0, cl, xf, IMG, g1, b1, g2, b2),)
            if kind == "x3": return (ops.layer_tail(Qf, kvi, tc, 0, cl, xf, IMG3, g1, b1, g2, b2),)
            if kind == "nq":
                qn = torch.empty_like(xf)
                return (ops.layer_tail(Qf, kvi, tc, 0, cl, xf, IMG_Q, g1, b1, g2, b2, q_next=qn), qn)
            if kind == "proj": return ops.proj_qkv(xf, PP, tc, cr, cl, 0, a_exp=A_EXP, k_exp=EXK[0], v_exp=EXK[1])
        r1 = run()
        y1c = r1[0].clone() if os.environ.get("QF_CLONE") == "1" else None
        corun()
        r2 = run()
        if y1c is not None and not torch.equal(y1c, r1[0]):
            print("  the first launch's y CHANGED after its clone was taken: %d elements" % int((y1c != r1[0]).sum()), flush=True)
        if y1c is not None and shown < 3 and not torch.equal(r1[0], r2[0]):
            r3 = run()
            print("  third launch equals: first %s, second %s" % (bool(torch.equal(r3[0], r1[0])), bool(torch.equal(r3[0], r2[0]))), flush=True)
        same = all(torch.equal(a, b) for a, b in zip(r1, r2))
        if not same and len(r1) == 2 and kind == "qf" and shown < 6:
            shown += 1
            ry = (ops.act_layout(r1[0], False) != ops.act_layout(r2[0], False)).any(dim=1).nonzero().flatten()
            dd = ops.act_layout(r1[1], False) != ops.act_layout(r2[1], False)
            rd = dd.any(dim=1).nonzero().flatten(); fd = dd.any(dim=0).nonzero().flatten().cpu().numpy()
            ya, yb = ops.act_layout(r1[0], False), ops.act_layout(r2[0], False)
            dy = (ya != yb)
            rows = dy.any(dim=1).nonzero().flatten().cpu().numpy(); feats = dy.any(dim=0).nonzero().flatten().cpu().numpy()
            print("  |dy| max %.3g; rows %d in %d row groups of 32 (whole groups: %s); features %d; per differing row, features that differ: min %d max %d; first groups %s"
                  % (float((ya - yb).abs().max()), len(rows), len(np.unique(rows // 32)), bool(len(rows) == 32 * len(np.unique(rows // 32))), len(feats),
                     int(dy[rows].sum(dim=1).min()), int(dy[rows].sum(dim=1).max()), np.unique(rows // 32)[:8]), flush=True)
            if os.environ.get("QF_ROWS") == "1":
                r = int(rows[0]); f = feats[:6]
                d1 = (ya[r] - yb[r]).double(); bb = b2.double(); gg = g2.double()
                # is the difference a row-wise scale (dy proportional to y - beta2) or an offset?
                t = ((ya[r].double() - bb)); k = float((d1 * t).sum() / (t * t).sum())
                resid = float((d1 - k * t).abs().max())
                print("  row %d: y1 %s\n           y2 %s\n  dy = %.3g * (y - beta2) + residual %.3g (max |dy| %.3g); rows of the group share the factor: %s"
                      % (r, ya[r, f].cpu().numpy(), yb[r, f].cpu().numpy(), k, resid, float(d1.abs().max()),
                         [round(float(((ya[q] - yb[q]).double() * (ya[q].double() - bb)).sum() / ((ya[q].double() - bb) ** 2).sum()), 10) for q in range(r - r % 32, r - r % 32 + 4)]), flush=True)
            if os.environ.get("QF_DUMP6") == "1":  # words 0-4 of every lane's 8: xor(apA), Zs(head 0), xor(apB), Zs(head 1), xor(Q' heads 0-6)
                a, b = r1[1].view(-1, 8), r2[1].view(-1, 8)
                ne = (a != b)
                print("  y differs in %d rows; checksum words that differ anywhere: %s (0 mean of norm2, 1 rstd, 2 var, 3 xor of the FFN accumulators in front of norm2, 4 y as computed, 5 the lane's partial sum, 6 the partner's as exchanged, 7 two exchanges agree); lanes %d"
                      % (len(ry), ne.any(dim=0).nonzero().flatten().tolist(), int(ne.any(dim=1).sum())), flush=True)
                continue
            print("  y differs in %d rows; the dumped intermediate in %d rows, %d features, chunks %s; rows of y not among them: %d"
                  % (len(ry), len(rd), len(fd), np.unique(fd // 32), len(set(ry.tolist()) - set(rd.tolist()))), flush=True)
    torch.cuda.synchronize()
    n += 1; bad += 0 if same else 1
print("%-5s beside %-5s: %d large draws in %.0f s, %d with two launches that differ" % (kind, mode, n, time.time() - t0, bad))
