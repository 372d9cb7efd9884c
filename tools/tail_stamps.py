#!/usr/bin/env python3
"""Cycles per stage of tail_kernel (T_SPLIT=h2 default / x3) from in-kernel s_memtime stamps (diagnostic build with -DT_STAMPS,
verified by tools/asm_inflight_check.py before it is linked; the shipped library has none).  `build` on the CPU box, `run` on the GPU: stamps of the second tile of every block, median over
blocks and waves, after a second of back-to-back launches so that the clock has settled."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_tabl", "t_stamps%s.so" % os.environ.get("T_TAG", ""))


def build():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_inflight_check as chk
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(ROOT, "scream_amd/csrc/tail_split.hip")
    flags = ["-ffp-contract=off", "-DT_STAMPS", *os.environ.get("T_EXTRA", "").split()]
    want = "11tail_kernelINS_" + {"h2": "7SplitH2ELb0ELb0", "x3": "8SplitBf3ELb0ELb0"}[os.environ.get("T_SPLIT", "h2")]  # the instance `run` launches
    assert chk.verify_source(src, flags, SO[:-3] + ".s", want) >= 1  # raises if a stamp pushed a pending register around
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags, src, "-o", SO])


def run():
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from scream_amd import _lib, ops, scales
    dev = "cuda:0"
    M = int(os.environ.get("T_M", 333312))
    split = {"h2": ops.SPLIT_H2, "x3": ops.SPLIT_BF3}[os.environ.get("T_SPLIT", "h2")]
    g = torch.Generator(device=dev).manual_seed(0)
    XMAX = 6.0
    x = torch.randn(M, 256, device=dev, generator=g).clamp_(-XMAX, XMAX)
    W1 = torch.randn(1024, 256, device=dev, generator=g) / 16; W2 = torch.randn(256, 1024, device=dev, generator=g) / 32
    Wm = torch.randn(256, 256, device=dev, generator=g) / 16; Wqkv = torch.randn(768, 256, device=dev, generator=g) / 16
    gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    n_tiles = M // 128; tpc = 40; n_clouds = (n_tiles + tpc - 1) // tpc
    tile_cloud = (torch.arange(n_tiles, device=dev) // tpc).int()
    crow0 = (torch.arange(n_clouds, device=dev) * tpc * 128).int()
    clen = torch.full((n_clouds,), tpc * 128 - 17, device=dev, dtype=torch.int32); clen[-1] = M - int(crow0[-1]) - 5
    xf = ops.act_layout(x, True)
    Qf, part = ops.gemm_qkv(xf, ops.pack_w(Wqkv, split), 256, tile_cloud, crow0, clen, 0, 3, a_exp=scales.exp_for(XMAX))
    Wv = torch.cat([Wqkv[384:512], Wqkv[640:768]])
    EX = ops.tail_exps(**scales.tail_exps(Wm, W1, W2, gam, bet, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wqkv[:256].abs().sum(dim=1).max())))
    lib = ctypes.CDLL(SO)
    V, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    ft = lib.scream_layer_tail_f32; ft.restype = ctypes.c_int
    ft.argtypes = [V, V, V, I32, V, V, V, V, V, V, V, V, V, I64, I32, ctypes.POINTER(_lib.TailExpsT), V]
    pt = lib.scream_pack_tail; pt.restype = ctypes.c_int; pt.argtypes = [V, V, V, V, I32, I32, ctypes.POINTER(_lib.TailExpsT), V, V]
    kf = lib.scream_kv_finalize_image; kf.restype = ctypes.c_int; kf.argtypes = [V, V, V, I64, I32, I32, V, I32, I64, I64, I32, V]
    tb = lib.scream_tail_image_bytes; tb.restype = ctypes.c_int64; tb.argtypes = [I32, I32]
    st = torch.cuda.current_stream().cuda_stream
    timg = torch.empty(tb(split, 0), device=dev, dtype=torch.uint8)
    assert pt(Wm.data_ptr(), W1.data_ptr(), W2.data_ptr(), None, 0, split, ctypes.byref(EX), timg.data_ptr(), st) == 0
    kvi = torch.zeros(n_clouds, lib.scream_kv_image_bytes(), device=dev, dtype=torch.uint8)
    assert kf(part.data_ptr(), crow0.data_ptr(), clen.data_ptr(), 0, 0, n_clouds, kvi.data_ptr(), 1, 0, 0, split, st) == 0
    y = torch.empty(M, 256, device=dev)
    call = lambda: ft(Qf.data_ptr(), kvi.data_ptr(), tile_cloud.data_ptr(), 0, clen.data_ptr(), xf.data_ptr(), timg.data_ptr(), gam.data_ptr(),
                      bet.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), None, M, split, ctypes.byref(EX), st)
    import time
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(10): assert call() == 0
        torch.cuda.synchronize()
    SLOTS = 24
    buf = (ctypes.c_longlong * (256 * 4 * SLOTS))()
    assert lib.scream_tail_stamps_read(buf) == 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(256 * 4, SLOTS)
    a = a[a[:, 5] > 0]
    d = np.diff(a[:, :6], axis=1)
    med = np.median(d, axis=0)
    per_stage = 16 * (3 if split == ops.SPLIT_H2 else 6) * 32  # MFMA cycles of a stage
    names = ["merge phase (8 stages, apply rides; floor 8 x %d)" % (per_stage + 384), "norm1 + split into planes",
             "FFN phase (64 stages; floor 64 x %d)" % per_stage, "drain + open apply of the next tile's head 1", "norm2 + y stores"]
    tot = np.median(a[:, 5] - a[:, 0])
    print("waves with stamps: %d; tile total median %d cycles (MFMA floor 72 x %d + 8 x 384 = %d)" % (a.shape[0], tot, per_stage, 72 * per_stage + 8 * 384))
    for nm, v in zip(names, med):
        print("  %-58s %8d  %5.1f %%" % (nm, v, 100.0 * v / tot))
    # stage tops inside the merge phase (low 32 bits of s_memtime, kept in scalar registers until the tile's end)
    m = a[:, 8:16] & 0xFFFFFFFF
    t0 = a[:, 0] & 0xFFFFFFFF
    t1 = a[:, 1] & 0xFFFFFFFF
    edges = np.concatenate([m, t1[:, None]], axis=1)
    dm = (np.diff(edges, axis=1)) & 0xFFFFFFFF
    print("  merge stages (top to top; the last one to the end of the phase): " + " ".join("%d" % v for v in np.median(dm, axis=0)))
    print("  tile start to the top of merge stage 0: %d" % np.median((m[:, 0] - t0) & 0xFFFFFFFF))
    # FFN: marks 10, 8 = tops of up(30), up(31); 11, 9 = tops of down(30), down(31).  Order: up30 down29 up31 down30 down31
    f = a[:, 16:20] & 0xFFFFFFFF  # marks 8, 9, 10, 11
    pair = (f[:, 0] - f[:, 2]) & 0xFFFFFFFF
    up = (f[:, 3] - f[:, 0]) & 0xFFFFFFFF
    dn30 = (f[:, 1] - f[:, 3]) & 0xFFFFFFFF
    print("  FFN: up stage %d, down stage %d (up + down pair %d); down(30), which also requests the next tile's head 0: %d"
          % (np.median(up), np.median(pair - up), np.median(pair), np.median(dn30)))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
