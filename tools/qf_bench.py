#!/usr/bin/env python3
"""A/B of one self layer with and without the layer tail's own query projection (tail_kernel<.., QF>), kernel by kernel on the stem
shape (M = 333 184 rows; T_M overrides): ring projection q | k | v (N = 768) + tail reading Q'  against  ring projection k | v only
(N = 512) + tail computing Q' = elu(x Wq^T) + 1 itself -- ms, sclk, socket power and joules per launch, and the sum per layer."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scream_amd import ops, scales
dev = "cuda:0"
secs = float(os.environ.get("T_SECS", 1.5))
M = int(os.environ.get("T_M", 333184))
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
        m = re.search(r"\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,.*,(\d+\.\d+)\s*$", out.strip().splitlines()[-1])
        if m: samples.append((time.time(), int(m.group(3)), float(m.group(4))))
        time.sleep(0.05)


threading.Thread(target=sampler, daemon=True).start()
g = torch.Generator(device=dev).manual_seed(0)
XMAX = 6.0
x = torch.randn(M, 256, device=dev, generator=g).clamp_(-XMAX, XMAX)
Wqkv = torch.randn(768, 256, device=dev, generator=g) / 16; Wm = torch.randn(256, 256, device=dev, generator=g) / 16
W1 = torch.randn(1024, 256, device=dev, generator=g) / 16; W2 = torch.randn(256, 1024, device=dev, generator=g) / 32
gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
nt = M // 128
tile_cloud = (torch.arange(nt, device=dev) // 40).int()
ncl = int(tile_cloud.max().item()) + 1
crow0 = (torch.arange(ncl, device=dev) * 40 * 128).int()
clen = torch.full((ncl,), 40 * 128 - 17, device=dev, dtype=torch.int32); clen[-1] = M - int(crow0[-1]) - 5
A_EXP = scales.exp_for(XMAX)
Wq, Wkv = Wqkv[:256], Wqkv[256:]
Wv = torch.cat([Wqkv[384:512], Wqkv[640:768]])
exd = scales.tail_exps(Wm, W1, W2, gam, bet, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wq.abs().sum(dim=1).max()))
rl = Wkv.abs().sum(dim=1).cpu().view(-1, 2, 128)
kw = dict(a_exp=A_EXP, k_exp=scales.exp_for(1.0 + XMAX * float(rl[:, 0].max())), v_exp=scales.exp_for(XMAX * float(rl[:, 1].max())))
xf = ops.act_layout(x, True)
P3 = ops.pack_proj(Wqkv, 256, ops.SPLIT_H2, scales.w_exp(Wqkv))
P2 = ops.pack_proj(Wkv.contiguous(), 0, ops.SPLIT_H2, scales.w_exp(Wkv))
Qf, part = ops.proj_qkv(xf, P3, tile_cloud, crow0, clen, 0, **kw)
kvi = ops.kv_finalize_image(part, crow0, clen, 0, 0, ncl, ncl, split=ops.SPLIT_H2)
plain = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, ops.tail_exps(**exd))
own = ops.pack_tail(Wm, W1, W2, ops.SPLIT_H2, ops.tail_exps(e_x=A_EXP, e_wq=scales.w_exp(Wq), **exd), Wq_own=Wq.contiguous())
y = torch.empty(M, 256, device=dev)
calls = [("proj q|k|v (N = 768)", lambda: ops.proj_qkv(xf, P3, tile_cloud, crow0, clen, 0, **kw)),
         ("proj k|v (N = 512)", lambda: ops.proj_qkv(xf, P2, tile_cloud, crow0, clen, 0, **kw)),
         ("tail reading Q'", lambda: ops.layer_tail(Qf, kvi, tile_cloud, 0, clen, xf, plain, gam, bet, gam, bet, out=y)),
         ("tail with its own Q projection", lambda: ops.layer_tail(None, kvi, tile_cloud, 0, clen, xf, own, gam, bet, gam, bet, out=y))]
y1 = ops.layer_tail(Qf, kvi, tile_cloud, 0, clen, xf, plain, gam, bet, gam, bet).clone()
y2 = ops.layer_tail(None, kvi, tile_cloud, 0, clen, xf, own, gam, bet, gam, bet)
print("# max |y(own Q) - y(read Q')| = %.3g" % float((y1 - y2).abs().max()))
print("%-34s %8s %9s %9s %9s" % ("kernel", "ms", "sclk MHz", "power W", "J/launch"), flush=True)
res = {}
for rep in range(int(os.environ.get("T_REPS", 2))):
    for name, f in calls:
        for _ in range(5): f()
        torch.cuda.synchronize(); time.sleep(0.3)
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(10): f()
            torch.cuda.synchronize(); n += 10
        t1 = time.time()
        sm = [s for s in samples if t0 + 0.4 <= s[0] <= t1]
        clk = sum(s[1] for s in sm) / max(len(sm), 1); pw = sum(s[2] for s in sm) / max(len(sm), 1)
        ms = (t1 - t0) / n * 1e3
        res[name] = (ms, pw * ms / 1e3)
        print("%-34s %8.3f %9.0f %9.0f %9.3f" % (name, ms, clk, pw, pw * ms / 1e3), flush=True)
    a = res["proj q|k|v (N = 768)"], res["tail reading Q'"]; b = res["proj k|v (N = 512)"], res["tail with its own Q projection"]
    print("# per layer: %.3f ms / %.3f J  ->  %.3f ms / %.3f J" % (a[0][0] + a[1][0], a[0][1] + a[1][1], b[0][0] + b[1][0], b[0][1] + b[1][1]), flush=True)
stop[0] = True
