#!/usr/bin/env python3
"""Energy per launch of the ablation variants of gemm_split_kernel (X3_SPLIT=h2 | x3) (tools/gemm_ablate.py build first): rocm-smi power and
sclk sampled while each variant runs for ~2.5 s on the FFN 256->1024 shape.  Under the 1400 W cap, time follows energy."""
import ctypes, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scream_amd import ops
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_ablate
import _split_ctypes as SC
dev = "cuda:0"; M = 327680
shape = os.environ.get("X3_SHAPE", "ffn1")
N, K, epi = {"ffn1": (1024, 256, ops.EPI_RELU), "ffn2": (256, 1024, ops.EPI_RES_LN), "qkv": (768, 256, ops.EPI_ELU1)}[shape]
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).clamp_(-8, 8); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
o = torch.empty(M, N, device=dev); r = torch.randn(M, 256, device=dev, generator=g); gam = torch.ones(256, device=dev)
V, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
        m = re.search(r"\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,.*,(\d+\.\d+)\s*$", out.strip().splitlines()[-1])
        if m: samples.append((time.time(), int(m.group(3)), float(m.group(4))))
        time.sleep(0.05)
th = threading.Thread(target=sampler, daemon=True); th.start()  # (daemon: a traceback in the main thread must end the process)
tag = os.environ.get("X3_TAG", "")
print("%-28s %9s %9s %9s %10s   (%s, M=%d; idle power ~296 W)" % ("variant", "ms", "sclk MHz", "power W", "J/launch", shape, M))
for bits, label in gemm_ablate.VARIANTS:
    f = os.path.join(ROOT, "tools", "_abl" + tag, "x3_%d.so" % bits)
    if not os.path.exists(f): continue
    lib, pack, gemm = SC.bind(f)
    st = torch.cuda.current_stream().cuda_stream
    Wp, w_exp = pack(W, st)
    call = lambda: gemm(A, Wp, w_exp, o, M, N, K, epi, 512 if epi == ops.EPI_ELU1 else 0, r, gam, st)
    call(); torch.cuda.synchronize(); time.sleep(0.5)
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.5:
        for _ in range(20): call()
        torch.cuda.synchronize(); n += 20
    t1 = time.time()
    win = [s for s in samples if t0 + 0.6 < s[0] < t1 - 0.1]
    sclk = sum(s[1] for s in win) / max(len(win), 1); pw = sum(s[2] for s in win) / max(len(win), 1)
    ms = (t1 - t0) / n * 1e3
    print("%-28s %9.3f %9.0f %9.0f %10.3f" % (label, ms, sclk, pw, pw * ms * 1e-3), flush=True)
    time.sleep(0.7)
stop = True; th.join()
