#!/usr/bin/env python3
"""Ablation of the q/k/v PROJECTION as the forward launches it (gemm_split_kernel<SplitH2, EPI_QKV, AFRAG>: fragment-major x in,
transposed query tile stored fragment-major, fused K^T V reduce on the key/value tiles) -- time, sclk, socket power and joules
per launch of every -DX3_ABLATE variant of scream_amd/csrc/gemm_split.hip, on the two shapes of a step:
    stem     M = 333 184 rows, N = 768  (q | k,v heads 0-3 | k,v heads 4-7)
    crosskv  M = 166 912 rows, N = 3072 (the six cross layers' target-side key/value projections, no queries)
`build` on the CPU box (every variant goes through tools/asm_inflight_check.py first, as in gemm_ablate.py), `run` on the GPU.
P_SRC=<file> selects another source with the same entry points (a candidate kernel), P_TAG its output directory suffix."""
import ctypes, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "tools", "_abl_proj" + os.environ.get("P_TAG", ""))
SRC = os.path.join(ROOT, os.environ.get("P_SRC", "scream_amd/csrc/gemm_split.hip"))
KERNEL = "17gemm_split_kernelINS_7SplitH2"
VARIANTS = [(0, "full"), (64, "no epilogue stores"), (128, "no K^T V epilogue (k/v tiles)"), (256, "no query epilogue (elu + stores)"),
            (1, "no epilogue"), (1 | 4, "no epilogue, no A loads"), (1 | 32, "no epilogue, no split"), (1 | 4 | 32, "no epilogue, no A loads, no split"),
            (1 | 2, "no epilogue, no W DMA"), (1 | 16, "no epilogue, no LDS reads"), (1 | 8, "no epilogue, no MFMA"),
            (1 | 2 | 4 | 16 | 32, "MFMA only"), (2 | 4 | 8 | 16 | 32, "epilogue only")]
if os.environ.get("P_VARIANTS"):
    VARIANTS = [v for v in VARIANTS if str(v[0]) in os.environ["P_VARIANTS"].split(",")]
EXTRA = os.environ.get("P_EXTRA", "").split()


def build():
    import asm_inflight_check as chk
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for bits, label in VARIANTS:
        flags = ["-DX3_ABLATE=%d" % bits, *EXTRA]
        try:  # a switch that removes memory operations changes what the counted waits leave in flight: never launch unverified code
            assert chk.verify_source(SRC, flags, os.path.join(OUT, "p_%d.s" % bits), os.environ.get("P_KERNEL", KERNEL)) >= 1
        except RuntimeError as e:
            print("SKIPPED variant %d (%s): %s" % (bits, label, e), flush=True)
            if os.path.exists(os.path.join(OUT, "p_%d.so" % bits)): os.remove(os.path.join(OUT, "p_%d.so" % bits))
            continue
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags, SRC, "-o", os.path.join(OUT, "p_%d.so" % bits)]
        procs.append(subprocess.Popen(cmd))
        if len(procs) == 4:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0


def bind(path):
    V, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    lib = ctypes.CDLL(path)
    fn = lib.scream_gemm_qkv_split_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [V, I64, V, V, I64, I64, I32, I32, I32, V, V, V, I64, V, I32, I32, I32, I32, I32, I32, V]
    pk = lib.scream_pack_w_split
    pk.restype = ctypes.c_int
    pk.argtypes = [V, I32, I32, I32, I32, V, V]
    return lib, fn, pk


def run():
    sys.path.insert(0, ROOT)
    import torch
    from scream_amd import ops, scales
    dev = "cuda:0"
    secs = float(os.environ.get("P_SECS", 1.5))
    g = torch.Generator(device=dev).manual_seed(0)
    XMAX = 6.0
    st = torch.cuda.current_stream().cuda_stream
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
            m = re.search(r"\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,.*,(\d+\.\d+)\s*$", out.strip().splitlines()[-1])
            if m: samples.append((time.time(), int(m.group(3)), float(m.group(4))))
            time.sleep(0.05)
    th = threading.Thread(target=sampler, daemon=True); th.start()
    shapes = [("stem", 333184, 768, 256), ("crosskv", 166912, 3072, 0)]
    if os.environ.get("P_SHAPES"):
        shapes = [s for s in shapes if s[0] in os.environ["P_SHAPES"].split(",")]
    tags = os.environ.get("P_TAGS", os.environ.get("P_TAG", "")).split(",")
    for name, M, N, n_q in shapes:
        x = torch.randn(M, 256, device=dev, generator=g).clamp_(-XMAX, XMAX)
        W = torch.randn(N, 256, device=dev, generator=g) / 16
        xf = ops.act_layout(x, True)
        n_tiles = M // 128; tpc = 40; n_clouds = (n_tiles + tpc - 1) // tpc
        tile_cloud = (torch.arange(n_tiles, device=dev) // tpc).int()
        crow0 = (torch.arange(n_clouds, device=dev) * tpc * 128).int()
        clen = torch.full((n_clouds,), tpc * 128 - 17, device=dev, dtype=torch.int32); clen[-1] = M - int(crow0[-1]) - 5
        a_exp, w_exp = scales.exp_for(XMAX), scales.w_exp(W)
        rl = W.abs().sum(dim=1)[n_q:].view(-1, 2, 128)
        k_exp, v_exp = scales.exp_for(1.0 + XMAX * float(rl[:, 0].max())), scales.exp_for(XMAX * float(rl[:, 1].max()))
        L = (N - n_q) // 512
        Q = torch.empty(M, 256, device=dev) if n_q else None
        part = torch.empty(L, M // 128, 8, 33 * 32, device=dev)
        flops = 2.0 * M * N * 256
        print("%-52s %9s %9s %9s %10s %9s  (%s: M=%d N=%d, fp32-equivalent %.1f GFLOP)" % ("variant", "ms", "sclk MHz", "power W", "J/launch", "TFLOP/s", name, M, N, flops / 1e9))
        for tag in tags:
            for bits, label in VARIANTS:
                f = os.path.join(ROOT, "tools", "_abl_proj" + tag, "p_%d.so" % bits)
                if not os.path.exists(f): continue
                lib, fn, pk = bind(f)
                Wp = torch.empty(2 * 2 * N * 256, device=dev, dtype=torch.uint8)
                assert pk(W.data_ptr(), N, 256, 2, w_exp, Wp.data_ptr(), st) == 0
                layout = 1 | (2 if n_q else 0)
                call = lambda: fn(xf.data_ptr(), 256, Wp.data_ptr(), Q.data_ptr() if n_q else None, 256, M, N, 256, n_q, tile_cloud.data_ptr(), crow0.data_ptr(),
                                  clen.data_ptr(), 0, part.data_ptr(), layout, 2, a_exp, w_exp, k_exp, v_exp, st)
                assert call() == 0
                torch.cuda.synchronize(); time.sleep(0.3)
                t0 = time.time(); n = 0
                while time.time() - t0 < secs:
                    for _ in range(20): call()
                    torch.cuda.synchronize(); n += 20
                t1 = time.time()
                win = [s for s in samples if t0 + 0.5 < s[0] < t1 - 0.1]
                sclk = sum(s[1] for s in win) / max(len(win), 1); pw = sum(s[2] for s in win) / max(len(win), 1)
                ms = (t1 - t0) / n * 1e3
                print("%-52s %9.3f %9.0f %9.0f %10.3f %9.1f" % ((tag + " " if tag else "") + label, ms, sclk, pw, pw * ms * 1e-3, flops / ms / 1e9), flush=True)
                time.sleep(0.4)
        del x, xf, W, part, Q
    stop[0] = True; th.join()


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
