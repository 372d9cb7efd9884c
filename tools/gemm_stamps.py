#!/usr/bin/env python3
"""Diagnostic: per-block s_memtime stamps of gemm_f32_kernel (wave 0, lane 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scream_amd import _lib, ops
lib = _lib.load()
lib.scream_gemm_debug_buffer.argtypes = [ctypes.c_void_p]
lib.scream_gemm_debug_mode.argtypes = [ctypes.c_int]
lib.scream_gemm_debug_mode(int(os.environ.get("MODE", "0")))
M, N, K = 327680, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 16; out = torch.empty(M, N, device=dev)
nblk = (M // 128) * (N // 256)
dbg = torch.zeros(nblk, 40, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.gemm_f32(A, W, ops.EPI_RELU, out=out)
torch.cuda.synchronize()
lib.scream_gemm_debug_buffer(dbg.data_ptr())
ops.gemm_f32(A, W, ops.EPI_RELU, out=out)
torch.cuda.synchronize()
lib.scream_gemm_debug_buffer(None)
d = dbg.cpu().numpy().astype(np.int64)
KT = K // 32
t0 = d[:, 0]; hw = d[:, 1]
stamps = d[:, 2:2 + 2 + 2 * KT]   # [after prologue] + per tile (before barrier, after barrier) + end
base = t0.min()
pro = stamps[:, 0] - t0
comp = np.stack([stamps[:, 1 + 2 * t] - (stamps[:, 0] if t == 0 else stamps[:, 2 * t]) for t in range(KT)], 1)
barr = np.stack([stamps[:, 2 + 2 * t] - stamps[:, 1 + 2 * t] for t in range(KT)], 1)
epi = stamps[:, 1 + 2 * KT] - stamps[:, 2 * KT]
life = stamps[:, 1 + 2 * KT] - t0
end = stamps[:, 1 + 2 * KT]
print("MODE", os.environ.get("MODE", "0"), "blocks", nblk, "kernel span cycles", int(end.max() - base), " ideal (MFMA-bound)", nblk * 4 * KT * 128 * 64 // 1024)
pc = lambda x: "p10 %.0f  med %.0f  p90 %.0f  mean %.0f" % (np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.mean())
print("prologue      ", pc(pro))
print("tile compute  ", pc(comp.ravel()), " (128 MFMA = 8192 cycles alone)")
print("barrier wait  ", pc(barr.ravel()))
print("epilogue      ", pc(epi))
print("block lifetime", pc(life))
print("sum per block: compute %.0f barrier %.0f" % (comp.sum(1).mean(), barr.sum(1).mean()))
# co-residency: group by (xcc, cu, se, sh) from HW_ID
cu_key = ((hw >> 32) & 7) * 65536 + ((hw & 0xffff) >> 8)
order = np.argsort(t0)
# for a few CUs print the timeline of blocks
keys = np.unique(cu_key)
print("distinct CU keys", len(keys))
for k in keys[:2]:
    idx = np.where(cu_key == k)[0]
    idx = idx[np.argsort(t0[idx])]
    print("CU key", k, "blocks", len(idx))
    for b in idx[:8]:
        print("   start %8d  end %8d  simd/wave %04x  tiles-end:" % (t0[b] - base, stamps[b, 1 + 2 * KT] - base, hw[b] & 0xff),
              " ".join("%d" % (stamps[b, 2 + 2 * t] - base) for t in range(KT)))
