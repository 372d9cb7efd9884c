#!/usr/bin/env python3
"""A/B of the two q/k/v projection kernels on the two shapes of a step (GPU): the 8-wave GEMM (scream_gemm_qkv_split_f32,
csrc/gemm_split.hip) and the ring kernel (scream_proj_qkv_f32, csrc/proj_ring.hip) -- ms per launch, fp32-equivalent TFLOP/s,
sclk and socket power (rocm-smi sampled beside back-to-back launches), joules per launch; outputs compared first.
    stem     M = 333 184 rows, N = 768  (q | k,v heads 0-3 | k,v heads 4-7)
    crossq   M = 166 912 rows, N = 768  (a cross-stage self layer)
    crosskv  M = 166 912 rows, N = 3072 (the six cross layers' target-side key/value projections, no queries)
SCREAM_LIB=<other build> compares builds; P_SECS seconds per leg."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scream_amd import ops, scales

dev = "cuda:0"
secs = float(os.environ.get("P_SECS", 1.5))
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
        m = re.search(r"\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,.*,(\d+\.\d+)\s*$", out.strip().splitlines()[-1])
        if m: samples.append((time.time(), int(m.group(3)), float(m.group(4))))
        time.sleep(0.05)


if os.environ.get("P_NOSMI") != "1":  # (under rocprofv3 starting another program from this process is refused on the GPU boxes)
    threading.Thread(target=sampler, daemon=True).start()
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("stem", 333184, 768, 256), ("crossq", 166912, 768, 256), ("crosskv", 166912, 3072, 0)]
if os.environ.get("P_SHAPES"):
    shapes = [s for s in shapes if s[0] in os.environ["P_SHAPES"].split(",")]
which = os.environ.get("P_KERNELS", "gemm,ring").split(",")
print("%-8s %-5s %8s %9s %9s %9s %9s" % ("shape", "kern", "ms", "TFLOP/s", "sclk MHz", "power W", "J/launch"), flush=True)
for name, M, N, n_q in shapes:
    x = torch.randn(M, 256, device=dev, generator=g).clamp_(-6, 6)
    W = torch.randn(N, 256, device=dev, generator=g) / 16
    nt = M // 128
    tile_cloud = torch.arange(nt, device=dev, dtype=torch.int32) // 40  # clouds of 40 tiles, the last rows of each padding
    ncl = int(tile_cloud.max().item()) + 1
    crow0 = (torch.arange(ncl, device=dev, dtype=torch.int32) * 40 * 128)
    clen = torch.full((ncl,), 40 * 128 - 77, device=dev, dtype=torch.int32)
    clen[-1] = M - int(crow0[-1].item()) - 5
    w_exp, a_exp = scales.w_exp(W), scales.exp_for(6.0)
    rl = W.abs().sum(dim=1).cpu()[n_q:].view(-1, 2, 128)
    k_exp, v_exp = scales.exp_for(1.0 + 6.0 * float(rl[:, 0].max())), scales.exp_for(6.0 * float(rl[:, 1].max()))
    xf = ops.act_layout(x, True)
    Wp = ops.pack_w(W, ops.SPLIT_H2, w_exp)
    P = ops.pack_proj(W, n_q, ops.SPLIT_H2, w_exp)
    kern = {"gemm": lambda: ops.gemm_qkv(xf, Wp, n_q, tile_cloud, crow0, clen, 0, layout=ops.LAYOUT_A_FRAG | (ops.LAYOUT_C_FRAG if n_q else 0),
                                         a_exp=a_exp, k_exp=k_exp, v_exp=v_exp),
            "ring": lambda: ops.proj_qkv(xf, P, tile_cloud, crow0, clen, 0, a_exp=a_exp, k_exp=k_exp, v_exp=v_exp)}
    Qa, pa = kern["gemm"]()
    Qb, pb = kern["ring"]()
    torch.cuda.synchronize()
    dq = float((Qa - Qb).abs().max()) if n_q else 0.0
    kva = ops.kv_finalize(pa if pa.dim() == 3 else pa[0], crow0, clen, 0, 0, ncl, ncl)
    kvb = ops.kv_finalize(pb if pb.dim() == 3 else pb[0], crow0, clen, 0, 0, ncl, ncl)
    print("# %s: max |Q' gemm - Q' ring| = %.3g, K^T V max rel diff = %.3g" % (name, dq, float(((kva - kvb).abs() / (kva.abs() + 1e-3)).max())), flush=True)
    flop = 2.0 * M * N * 256
    for kn in which:
        f = kern[kn]
        for _ in range(5): f()
        torch.cuda.synchronize()
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < secs:
            for _ in range(20): f()
            n += 20
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        t1 = time.time()
        ms = e0.elapsed_time(e1) / n
        sm = [s for s in samples if t0 + 0.4 <= s[0] <= t1]
        clk = sum(s[1] for s in sm) / max(len(sm), 1); pw = sum(s[2] for s in sm) / max(len(sm), 1)
        print("%-8s %-5s %8.3f %9.1f %9.0f %9.0f %9.3f" % (name, kn, ms, flop / ms / 1e9, clk, pw, pw * ms / 1e3), flush=True)
stop[0] = True
