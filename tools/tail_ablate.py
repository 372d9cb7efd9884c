#!/usr/bin/env python3
"""Where the layer-tail kernel (scream_amd/csrc/tail_split.hip) spends its time and energy: builds the file with
-DT_ABLATE=<bits> into tools/_tabl/ (`build`, on the CPU box -- every variant goes through tools/asm_inflight_check.py first: a
switch that removes memory operations changes what the hand-counted waits leave in flight) and runs every variant for ~2 s on
M rows while rocm-smi samples socket power and sclk (`run`, on the GPU).  T_SPLIT=h2 (default) / x3 selects the operand split;
the unfused chain (attention apply + three split GEMMs) runs beside them as the yardstick."""
import ctypes, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_tabl" + os.environ.get("T_TAG", ""))
VARIANTS = [(0, "full"), (1, "no W DMA"), (16, "no row I/O"), (32, "no Q' loads"), (64, "no x loads"), (128, "no y stores"),
            (256, "no KV^T / Ksum loads"), (512, "no apply rides"), (4, "no LDS reads"), (1 | 4, "no DMA, no LDS reads"),
            (2, "no MFMA"), (1 | 4 | 16, "MFMA only")]
if os.environ.get("T_VARIANTS"):
    VARIANTS = [v for v in VARIANTS if str(v[0]) in os.environ["T_VARIANTS"].split(",")]
EXTRA = os.environ.get("T_EXTRA", "").split()
SRC = os.path.join(ROOT, os.environ.get("T_SRC", "scream_amd/csrc/tail_split.hip"))


def build():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_inflight_check as chk
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for bits, label in VARIANTS:
        flags = ["-ffp-contract=off", "-DT_ABLATE=%d" % bits, *EXTRA]
        want = "11tail_kernelINS_" + {"h2": "7SplitH2ELb0ELb0", "x3": "8SplitBf3ELb0ELb0"}[os.environ.get("T_SPLIT", "h2")]  # the instance `run` will launch
        try:
            assert chk.verify_source(SRC, flags, os.path.join(OUT, "t_%d.s" % bits), want) >= 1
        except RuntimeError as e:  # never launch a variant whose generated code touches a pending register: skip it, loudly
            print("SKIPPED variant %d (%s): %s" % (bits, label, e), flush=True)
            if os.path.exists(os.path.join(OUT, "t_%d.so" % bits)): os.remove(os.path.join(OUT, "t_%d.so" % bits))
            continue
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags, SRC, "-o", os.path.join(OUT, "t_%d.so" % bits)]
        procs.append(subprocess.Popen(cmd))
        if len(procs) == 4:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0


def run():
    sys.path.insert(0, ROOT)
    import torch
    from scream_amd import _lib, ops, scales
    dev = "cuda:0"
    M = int(os.environ.get("T_M", 333312))
    secs = float(os.environ.get("T_SECS", 2.0))
    split = {"h2": ops.SPLIT_H2, "x3": ops.SPLIT_BF3}[os.environ.get("T_SPLIT", "h2")]
    g = torch.Generator(device=dev).manual_seed(0)
    XMAX = 6.0
    x = torch.randn(M, 256, device=dev, generator=g).clamp_(-XMAX, XMAX)
    W1 = torch.randn(1024, 256, device=dev, generator=g) / 16; W2 = torch.randn(256, 1024, device=dev, generator=g) / 32
    Wqkv = torch.randn(768, 256, device=dev, generator=g) / 16; Wm = torch.randn(256, 256, device=dev, generator=g) / 16
    gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    y = torch.empty(M, 256, device=dev); hid = torch.empty(M, 1024, device=dev); att2 = torch.empty(M, 256, device=dev)
    V, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    st = torch.cuda.current_stream().cuda_stream
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
            m = re.search(r"\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,.*,(\d+\.\d+)\s*$", out.strip().splitlines()[-1])
            if m: samples.append((time.time(), int(m.group(3)), float(m.group(4))))
            time.sleep(0.05)
    th = threading.Thread(target=sampler, daemon=True); th.start()  # (daemon: a traceback in the main thread must end the process)
    # synthetic but well-formed operands (one cloud per 40 row tiles)
    n_tiles = M // 128
    tiles_per_cloud = 40
    n_clouds = (n_tiles + tiles_per_cloud - 1) // tiles_per_cloud
    tile_cloud = (torch.arange(n_tiles, device=dev) // tiles_per_cloud).int()
    crow0 = (torch.arange(n_clouds, device=dev) * tiles_per_cloud * 128).int()
    clen = torch.full((n_clouds,), tiles_per_cloud * 128 - 17, device=dev, dtype=torch.int32)
    clen[-1] = M - int(crow0[-1]) - 5
    A_EXP = scales.exp_for(XMAX)
    Wv = torch.cat([Wqkv[384:512], Wqkv[640:768]])
    EX = ops.tail_exps(**scales.tail_exps(Wm, W1, W2, gam, bet, XMAX * float(Wv.abs().sum(dim=1).max()), XMAX * float(Wqkv[:256].abs().sum(dim=1).max())))
    pq, pm, p1, p2 = (ops.pack_w(w, split) for w in (Wqkv, Wm, W1, W2))
    xf = ops.act_layout(x, True)
    Qf, part = ops.gemm_qkv(xf, pq, 256, tile_cloud, crow0, clen, 0, 3, a_exp=A_EXP)
    Qp = ops.act_layout(Qf, False)
    kv = ops.kv_finalize(part, crow0, clen, 0, 0, n_clouds, n_clouds)

    def unfused_tail():
        a = ops.attn_apply(Qp, 256, kv, tile_cloud, 0, clen, M)
        ops.gemm_split(a, pm, ops.EPI_RES_LN, residual=x, gamma=gam, beta=bet, out=att2, a_exp=EX.e_att)
        ops.gemm_split(att2, p1, ops.EPI_RELU, out=hid, a_exp=EX.e_m1)
        ops.gemm_split(hid, p2, ops.EPI_RES_LN, residual=x, gamma=gam, beta=bet, out=y, a_exp=EX.e_h)
    calls = [("apply + merge + FFN-up + FFN-down (4 launches)", unfused_tail)]
    for tag in os.environ.get("T_TAGS", "").split(","):
        for bits, label in VARIANTS:
            f = os.path.join(ROOT, "tools", "_tabl" + tag, "t_%d.so" % bits)
            if not os.path.exists(f): continue
            lib = ctypes.CDLL(f)
            ft = lib.scream_layer_tail_f32; ft.restype = ctypes.c_int
            ft.argtypes = [V, V, V, I32, V, V, V, V, V, V, V, V, V, I64, I32, ctypes.POINTER(_lib.TailExpsT), V]
            pt = lib.scream_pack_tail; pt.restype = ctypes.c_int; pt.argtypes = [V, V, V, V, I32, I32, ctypes.POINTER(_lib.TailExpsT), V, V]
            kf = lib.scream_kv_finalize_image; kf.restype = ctypes.c_int; kf.argtypes = [V, V, V, I64, I32, I32, V, I32, I64, I64, I32, V]
            tb = lib.scream_tail_image_bytes; tb.restype = ctypes.c_int64; tb.argtypes = [I32, I32]
            timg = torch.empty(tb(split, 0), device=dev, dtype=torch.uint8)
            assert pt(Wm.data_ptr(), W1.data_ptr(), W2.data_ptr(), None, 0, split, ctypes.byref(EX), timg.data_ptr(), st) == 0
            kvi = torch.zeros(n_clouds, lib.scream_kv_image_bytes(), device=dev, dtype=torch.uint8)
            assert kf(part.data_ptr(), crow0.data_ptr(), clen.data_ptr(), 0, 0, n_clouds, kvi.data_ptr(), 1, 0, 0, split, st) == 0
            calls.append(((tag + " " if tag else "") + "fused tail: " + label,
                          (lambda ft=ft, timg=timg, kvi=kvi: ft(Qf.data_ptr(), kvi.data_ptr(), tile_cloud.data_ptr(), 0, clen.data_ptr(), xf.data_ptr(),
                                                                timg.data_ptr(), gam.data_ptr(), bet.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), None, M,
                                                                split, ctypes.byref(EX), st))))
    print("%-50s %9s %9s %9s %10s %9s  (M=%d, split %s; merge + FFN = %.1f GFLOP)" % ("variant", "ms", "sclk MHz", "power W", "J/launch", "TFLOP/s", M, os.environ.get("T_SPLIT", "h2"), 2.0 * M * 256 * 2304 / 1e9))
    y_ref = None
    for label, call in calls:
        call(); torch.cuda.synchronize(); time.sleep(0.4)
        if "fused tail: full" in label:  # every build of the whole kernel must give the same bits (T_TAGS: A/B of code variants)
            if y_ref is None: y_ref = y.clone()
            else: print("# %s: max |y - y of the first full build| = %g" % (label, float((y - y_ref).abs().max())), flush=True)
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(10): call()
            torch.cuda.synchronize(); n += 10
        t1 = time.time()
        win = [s for s in samples if t0 + 0.5 < s[0] < t1 - 0.1]
        sclk = sum(s[1] for s in win) / max(len(win), 1); pw = sum(s[2] for s in win) / max(len(win), 1)
        ms = (t1 - t0) / n * 1e3
        print("%-50s %9.3f %9.0f %9.0f %10.3f %9.1f" % (label, ms, sclk, pw, pw * ms * 1e-3, 2.0 * M * 256 * 2304 / ms / 1e9), flush=True)
        time.sleep(0.5)
    stop[0] = True; th.join()


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
