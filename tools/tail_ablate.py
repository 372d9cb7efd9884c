#!/usr/bin/env python3
"""Where the fused FFN kernel (scream_amd/csrc/tail_x3.hip) spends its time and energy: builds the file with
-DT_ABLATE=<bits> into tools/_tabl/ (`build`, on the CPU box) and runs every variant for ~2 s on M rows while rocm-smi
samples socket power and sclk (`run`, on the GPU).  The two-launch FFN of gemm_x3.hip runs beside them as the yardstick."""
import ctypes, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_tabl" + os.environ.get("T_TAG", ""))
VARIANTS = [(0, "full"), (1, "no W DMA"), (16, "no row I/O"), (32, "no Q' loads"), (64, "no x loads"), (128, "no y stores"),
            (256, "no KV^T / Ksum loads"), (8, "no relu/split"), (4, "no LDS reads"), (1 | 4, "no DMA, no LDS reads"),
            (2, "no MFMA"), (1 | 4 | 8 | 16, "MFMA only")]
if os.environ.get("T_VARIANTS"):
    VARIANTS = [v for v in VARIANTS if str(v[0]) in os.environ["T_VARIANTS"].split(",")]
EXTRA = os.environ.get("T_EXTRA", "").split()


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for bits, _ in VARIANTS:
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-DT_ABLATE=%d" % bits, *EXTRA,
               os.path.join(ROOT, os.environ.get("T_SRC", "scream_amd/csrc/tail_x3.hip")), "-o", os.path.join(OUT, "t_%d.so" % bits)]
        procs.append(subprocess.Popen(cmd))
        if len(procs) == 4:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0


def run():
    sys.path.insert(0, ROOT)
    import torch
    from scream_amd import ops
    dev = "cuda:0"
    M = int(os.environ.get("T_M", 333312))
    secs = float(os.environ.get("T_SECS", 2.0))
    g = torch.Generator(device=dev).manual_seed(0)
    m1 = torch.randn(M, 256, device=dev, generator=g); x = torch.randn(M, 256, device=dev, generator=g)
    W1 = torch.randn(1024, 256, device=dev, generator=g) / 16; W2 = torch.randn(256, 1024, device=dev, generator=g) / 32
    gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    y = torch.empty(M, 256, device=dev); hid = torch.empty(M, 1024, device=dev)
    p1, p2 = ops.split_planes(W1), ops.split_planes(W2)
    V, I64 = ctypes.c_void_p, ctypes.c_int64
    st = torch.cuda.current_stream().cuda_stream
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
            m = re.search(r"\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,.*,(\d+\.\d+)\s*$", out.strip().splitlines()[-1])
            if m: samples.append((time.time(), int(m.group(3)), float(m.group(4))))
            time.sleep(0.05)
    th = threading.Thread(target=sampler); th.start()

    def two():
        ops.gemm_x3(m1, p1, ops.EPI_RELU, out=hid)
        ops.gemm_x3(hid, p2, ops.EPI_RES_LN, residual=x, gamma=gam, beta=bet, out=y)
    # the whole layer tail: synthetic but well-formed operands (one cloud per 40 row tiles)
    n_tiles = M // 128
    tiles_per_cloud = 40
    n_clouds = (n_tiles + tiles_per_cloud - 1) // tiles_per_cloud
    tile_cloud = (torch.arange(n_tiles, device=dev) // tiles_per_cloud).int()
    crow0 = (torch.arange(n_clouds, device=dev) * tiles_per_cloud * 128).int()
    clen = torch.full((n_clouds,), tiles_per_cloud * 128 - 17, device=dev, dtype=torch.int32)
    clen[-1] = M - int(crow0[-1]) - 5
    Wqkv = torch.randn(768, 256, device=dev, generator=g) / 16
    Wm = torch.randn(256, 256, device=dev, generator=g) / 16
    Qp, part = ops.gemm_qkv(x, ops.split_planes(Wqkv), 256, tile_cloud, crow0, clen, 0)
    Qf, xf = ops.act_layout(Qp, True), ops.act_layout(x, True)  # the fused tail takes fragment-major operands
    kv = ops.kv_finalize(part, crow0, clen, 0, 0, n_clouds, n_clouds)
    att = torch.empty(M, 256, device=dev)
    pm = ops.split_planes(Wm)
    ffn_img = ops.pack_ffn(W1, W2)

    def unfused_tail():
        a = ops.attn_apply(Qp, 256, kv, tile_cloud, 0, clen, M)
        ops.gemm_x3(a, pm, ops.EPI_RES_LN, residual=x, gamma=gam, beta=bet, out=hid[:, :256].contiguous() if False else att)
        ops.ffn_x3(att, ffn_img, x, gam, bet, out=y)
    calls = [("two launches (gemm_x3 up + down)", two), ("apply + merge GEMM + fused FFN (3 launches)", unfused_tail)]
    for tag in os.environ.get("T_TAGS", "").split(","):
        for bits, label in VARIANTS:
            f = os.path.join(ROOT, "tools", "_tabl" + tag, "t_%d.so" % bits)
            if not os.path.exists(f): continue
            lib = ctypes.CDLL(f)
            fn = lib.scream_ffn_x3_f32; fn.restype = ctypes.c_int; fn.argtypes = [V, I64, V, V, I64, V, V, V, I64, I64, V]
            pk = lib.scream_pack_ffn_x3; pk.restype = ctypes.c_int; pk.argtypes = [V, V, V, V]
            img = torch.empty(lib.scream_ffn_image_bytes(), device=dev, dtype=torch.uint8)
            assert pk(W1.data_ptr(), W2.data_ptr(), img.data_ptr(), st) == 0
            if not os.environ.get("T_NO_FFN"):
                calls.append(((tag + " " if tag else "") + "fused FFN: " + label,
                              (lambda fn=fn, img=img: fn(m1.data_ptr(), 256, img.data_ptr(), x.data_ptr(), 256, gam.data_ptr(), bet.data_ptr(), y.data_ptr(), 256, M, st))))
            I32 = ctypes.c_int32
            ft = lib.scream_layer_tail_x3_f32; ft.restype = ctypes.c_int
            ft.argtypes = [V, V, V, I32, V, V, V, V, V, V, V, V, I64, V]
            pt = lib.scream_pack_tail_x3; pt.restype = ctypes.c_int; pt.argtypes = [V, V, V, V, V]
            kf = lib.scream_kv_finalize_x3; kf.restype = ctypes.c_int; kf.argtypes = [V, V, V, I64, I32, I32, V, V]
            timg = torch.empty(lib.scream_tail_image_bytes(), device=dev, dtype=torch.uint8)
            assert pt(Wm.data_ptr(), W1.data_ptr(), W2.data_ptr(), timg.data_ptr(), st) == 0
            kvi = torch.zeros(n_clouds, lib.scream_kv_image_bytes(), device=dev, dtype=torch.uint8)
            assert kf(part.data_ptr(), crow0.data_ptr(), clen.data_ptr(), 0, 0, n_clouds, kvi.data_ptr(), st) == 0
            calls.append(((tag + " " if tag else "") + "fused tail: " + label,
                          (lambda ft=ft, timg=timg, kvi=kvi: ft(Qf.data_ptr(), kvi.data_ptr(), tile_cloud.data_ptr(), 0, clen.data_ptr(), xf.data_ptr(),
                                                                timg.data_ptr(), gam.data_ptr(), bet.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), M, st))))
    print("%-46s %9s %9s %9s %10s %9s  (M=%d; FFN up+down = %.1f GFLOP)" % ("variant", "ms", "sclk MHz", "power W", "J/launch", "TFLOP/s", M, 4.0 * M * 256 * 1024 / 1e9))
    for label, call in calls:
        call(); torch.cuda.synchronize(); time.sleep(0.4)
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(10): call()
            torch.cuda.synchronize(); n += 10
        t1 = time.time()
        win = [s for s in samples if t0 + 0.5 < s[0] < t1 - 0.1]
        sclk = sum(s[1] for s in win) / max(len(win), 1); pw = sum(s[2] for s in win) / max(len(win), 1)
        ms = (t1 - t0) / n * 1e3
        fl = (2.0 * M * 256 * 2304 if ("tail" in label or "3 launches" in label) else 4.0 * M * 256 * 1024)
        print("%-46s %9.3f %9.0f %9.0f %10.3f %9.1f" % (label, ms, sclk, pw, pw * ms * 1e-3, fl / ms / 1e9), flush=True)
        time.sleep(0.5)
    stop[0] = True; th.join()


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
