#!/usr/bin/env python3
"""In-kernel s_memtime stamps of gemm_split_kernel (X3_SPLIT=h2 | x3; third output tile of every block): where a k-tile's time goes.
   build: hipcc ... -DX3_STAMPS -> tools/_abl_stamps/x3_0.so (python tools/gemm_stamps.py build);  run on the GPU."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_abl_stamps", "x3_0.so")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _split_ctypes as SC
if sys.argv[1:] == ["build"]:
    import asm_inflight_check as chk
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(ROOT, SC.SRC)
    assert chk.verify_source(src, ["-DX3_STAMPS"], SO[:-3] + ".s", SC.KERNEL) == 12  # never launch unverified counted waits
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DX3_STAMPS", src, "-o", SO])
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from scream_amd import ops
lib, pack, gemm = SC.bind(SO)
dev = "cuda:0"; M = 327680
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, epi in [("ffn1", 1024, 256, ops.EPI_RELU), ("merge", 256, 256, ops.EPI_RES_LN), ("ffn2", 256, 1024, ops.EPI_RES_LN)]:
    A = torch.randn(M, K, device=dev, generator=g).clamp_(-8, 8); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    o = torch.empty(M, N, device=dev); r = torch.randn(M, 256, device=dev, generator=g); gam = torch.ones(256, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    Wp, w_exp = pack(W, st)
    for _ in range(3):
        assert gemm(A, Wp, w_exp, o, M, N, K, epi, 0, r, gam, st) == 0
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 160, dtype=np.int64)
    assert lib.scream_gemm_stamps_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf.reshape(256, 8, 160).astype(np.float64)
    KT = K // 32
    t0 = s[:, :, 0:1]
    kt = s[:, :, 4:4 + 4 * KT].reshape(256, 8, KT, 4)
    wait = kt[..., 1] - kt[..., 0]            # barrier arrival -> release
    s0 = kt[..., 2] - kt[..., 1]              # 8 groups of step 0
    s1 = kt[..., 3] - kt[..., 2]              # requests + split + 8 groups of step 1
    nxt = np.concatenate([kt[:, :, 1:, 0], s[:, :, 1:2]], axis=2) - kt[..., 3]  # split of next step 0, to the next barrier arrival
    print("%s  K=%d: per k-tile, mean over blocks (cycles)   main loop %.0f  epilogue %.0f" % (name, K, (s[:, :, 1] - s[:, :, 0]).mean(), (s[:, :, 2] - s[:, :, 1]).mean()))
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        print("   %s: barrier wait %6.0f | step0 %6.0f | requests+split+step1 %6.0f | tail split %5.0f   (first k-tile wait %6.0f, last %6.0f)" % (
            grp, wait[:, sl, 1:].mean(), s0[:, sl].mean(), s1[:, sl].mean(), nxt[:, sl].mean(), wait[:, sl, 0].mean(), wait[:, sl, -1].mean()))
