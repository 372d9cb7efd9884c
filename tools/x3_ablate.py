#!/usr/bin/env python3
"""Ablation timing of gemm_x3_kernel: builds scream_amd/csrc/gemm_x3.hip with -DX3_ABLATE=<bits> into tools/_abl/
(`build`, on the CPU box) and times each variant on the forward's four GEMM shapes (`run`, on the GPU)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_abl" + os.environ.get("X3_TAG", ""))
VARIANTS = [(0, "full")] if os.environ.get("X3_ONLY_FULL") else [(0, "full"), (64, "no epilogue stores"), (1, "no epilogue"), (1 | 2, "no epilogue, no W DMA"), (1 | 4, "no epilogue, no A loads"),
            (1 | 32, "no epilogue, no split"), (1 | 16, "no epilogue, no LDS reads"), (1 | 8, "no epilogue, no MFMA"),
            (1 | 2 | 4 | 16 | 32, "MFMA only"), (2 | 4 | 8 | 16 | 32, "epilogue only")]
if os.environ.get("X3_VARIANTS"):  # e.g. X3_VARIANTS=0,1 -> only those ablation bit sets
    VARIANTS = [v for v in VARIANTS if str(v[0]) in os.environ["X3_VARIANTS"].split(",")]
EXTRA = os.environ.get("X3_EXTRA", "").split()

def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for bits, _ in VARIANTS:
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DX3_ABLATE=%d" % bits, *EXTRA,
               os.path.join(ROOT, os.environ.get("X3_SRC", "scream_amd/csrc/gemm_x3.hip")), "-o", os.path.join(OUT, "x3_%d.so" % bits)]
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
            lib = ctypes.CDLL(f)
            fn = lib.scream_gemm_x3_f32
            fn.restype = ctypes.c_int
            fn.argtypes = [V, I64, V, V, I64, I64, I32, I32, I32, I32, V, V, I64, V, V, V]
            pk = lib.scream_pack_w_x3
            pk.restype = ctypes.c_int
            pk.argtypes = [V, I32, I32, V, V]
            libs.append(((tag or "-") + " " + label, fn, pk))
    print("%-40s" % "variant" + "".join("%16s" % s[0] for s in shapes) + "   (ms | fp32-equivalent TFLOP/s)")
    res = {name: {} for name, *_ in shapes}
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        o = torch.empty(M, N, device=dev); rsd = torch.randn(M, 256, device=dev, generator=g); gam = torch.ones(256, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        calls = []
        for label, fn, pk in libs:  # every build packs with its own packer (the image layout belongs to the kernel)
            Wp = torch.empty(6 * N * K, device=dev, dtype=torch.uint8)
            assert pk(W.data_ptr(), N, K, Wp.data_ptr(), st) == 0
            calls.append((label, Wp, (lambda fn=fn, Wp=Wp: fn(A.data_ptr(), K, Wp.data_ptr(), o.data_ptr(), N, M, N, K, epi, 512 if epi == ops.EPI_ELU1 else 0,
                                                                 None, rsd.data_ptr(), 256, gam.data_ptr(), gam.data_ptr(), st))))
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
