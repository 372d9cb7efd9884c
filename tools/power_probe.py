#!/usr/bin/env python3
"""Sample rocm-smi (power, sclk) while the x3 / f32 GEMM runs back to back: is the split GEMM power-limited?"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import ops
dev = "cuda:0"; M = 327680
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, 256, device=dev, generator=g); W = torch.randn(1024, 256, device=dev, generator=g) / 16
Wp = ops.pack_w(W); o = torch.empty(M, 1024, device=dev)
samples, stop = [], False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.time(), "ERR %r" % (e,)))
        time.sleep(0.05)
th = threading.Thread(target=sampler, daemon=True); th.start()  # (daemon: a traceback in the main thread must end the process)
def burn(fn, secs, tag):
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
    print("%s: %.3f ms per launch, %.1f TFLOP/s, window [%.2f, %.2f]" % (tag, dt / n * 1e3, 2.0 * M * 1024 * 256 * n / dt / 1e12, t0, t0 + dt), flush=True)
time.sleep(1.0)
burn(lambda: ops.gemm_split(A, Wp, ops.EPI_RELU, out=o), 4.0, "x3  FFN 256->1024")
time.sleep(1.0)
burn(lambda: ops.gemm_f32(A, W, ops.EPI_RELU, out=o), 4.0, "f32 FFN 256->1024")
time.sleep(1.0)
stop = True; th.join()
print("samples:", len(samples))
for t, s in samples[:: max(1, len(samples) // 60)]:
    print("%.2f %s" % (t, s[:400]))
