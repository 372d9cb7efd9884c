#!/usr/bin/env python3
"""Ablation timing of gemm_split_kernel (X3_SPLIT=h2 default | x3): builds scream_amd/csrc/gemm_split.hip with
-DX3_ABLATE=<bits> into tools/_abl/ (`build`, on the CPU box; every variant passes tools/asm_inflight_check.py first or is
skipped) and times each variant on the forward's four GEMM shapes (`run`, on the GPU)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _split_ctypes as SC
OUT = os.path.join(ROOT, "tools", "_abl" + os.environ.get("X3_TAG", ""))
VARIANTS = [(0, "full")] if os.environ.get("X3_ONLY_FULL") else [(0, "full"), (64, "no epilogue stores"), (1, "no epilogue"), (1 | 2, "no epilogue, no W DMA"), (1 | 4, "no epilogue, no A loads"),
            (1 | 32, "no epilogue, no split"), (1 | 16, "no epilogue, no LDS reads"), (1 | 8, "no epilogue, no MFMA"),
            (1 | 2 | 4 | 16 | 32, "MFMA only"), (2 | 4 | 8 | 16 | 32, "epilogue only")]
if os.environ.get("X3_VARIANTS"):  # e.g. X3_VARIANTS=0,1 -> only those ablation bit sets
    VARIANTS = [v for v in VARIANTS if str(v[0]) in os.environ["X3_VARIANTS"].split(",")]
EXTRA = os.environ.get("X3_EXTRA", "").split()

def build():
    import asm_inflight_check as chk
    os.makedirs(OUT, exist_ok=True)
    procs = []
    src = os.path.join(ROOT, os.environ.get("X3_SRC", SC.SRC))
    for bits, label in VARIANTS:
        flags = ["-DX3_ABLATE=%d" % bits, *EXTRA]
        try:  # a switch that removes memory operations changes what the counted waits leave in flight: never launch unverified code
            assert chk.verify_source(src, flags, os.path.join(OUT, "x3_%d.s" % bits), SC.KERNEL) == 12
        except RuntimeError as e:
            print("SKIPPED variant %d (%s): %s" % (bits, label, e), flush=True)
            if os.path.exists(os.path.join(OUT, "x3_%d.so" % bits)): os.remove(os.path.join(OUT, "x3_%d.so" % bits))
            continue
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags, src, "-o", os.path.join(OUT, "x3_%d.so" % bits)]
        procs.append(subprocess.Popen(cmd))
        if len(procs) == 4:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0

def run():
    """X3_TAGS=tag1,tag2 (default: the untagged build) x every variant present; variants are timed round-robin in one
    process (7 rounds of 5 launches per shape, median) so that clock and cache state are shared."""
    sys.path.insert(0, ROOT)
    import torch
    from scream_amd import ops
    dev = "cuda:0"
    M = int(os.environ.get("X3_M", 327680))
    V, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [("qkv", 768, 256, ops.EPI_ELU1), ("ffn1", 1024, 256, ops.EPI_RELU), ("ffn2", 256, 1024, ops.EPI_RES_LN), ("merge", 256, 256, ops.EPI_RES_LN)]
    tags = os.environ.get("X3_TAGS", "").split(",")
    libs = []
    for tag in tags:
        d = os.path.join(ROOT, "tools", "_abl" + tag)
        for bits, label in VARIANTS:
            f = os.path.join(d, "x3_%d.so" % bits)
            if not os.path.exists(f):
                continue
            lib, pack, gemm = SC.bind(f)
            libs.append(((tag or "-") + " " + label, gemm, pack))
    print("%-40s" % "variant" + "".join("%16s" % s[0] for s in shapes) + "   (ms | fp32-equivalent TFLOP/s)")
    res = {name: {} for name, *_ in shapes}
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev, generator=g).clamp_(-8, 8); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        o = torch.empty(M, N, device=dev); rsd = torch.randn(M, 256, device=dev, generator=g); gam = torch.ones(256, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        calls = []
        for label, fn, pk in libs:  # every build packs with its own packer (the image layout belongs to the kernel)
            Wp, w_exp = pk(W, st)
            calls.append((label, Wp, (lambda fn=fn, Wp=Wp, w_exp=w_exp: fn(A, Wp, w_exp, o, M, N, K, epi, 512 if epi == ops.EPI_ELU1 else 0, rsd, gam, st))))
        for label, _, call in calls:
            assert call() == 0
        torch.cuda.synchronize()
        ts = {label: [] for label, _, _ in calls}
        for rnd in range(7):
            for label, _, call in calls:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): call()
                e1.record(); torch.cuda.synchronize(); ts[label].append(e0.elapsed_time(e1) / 5)
        for label in ts:
            t = sorted(ts[label])[3]
            res[name][label] = "  %6.3f | %5.1f" % (t, 2.0 * M * N * K / t / 1e9)
        del A, W, o, rsd
    for label, _, _ in libs:
        print("%-40s" % label + "".join(res[name][label] for name, *_ in shapes), flush=True)

if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
