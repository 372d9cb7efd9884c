#!/usr/bin/env python3
"""Ablation timing of gemm_x3_kernel: builds scream_amd/csrc/gemm_x3.hip with -DX3_ABLATE=<bits> into tools/_abl/
(`build`, on the CPU box) and times each variant on the forward's four GEMM shapes (`run`, on the GPU)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_abl" + os.environ.get("X3_TAG", ""))
VARIANTS = [(0, "full")] if os.environ.get("X3_ONLY_FULL") else [(0, "full"), (64, "no epilogue stores"), (1, "no epilogue"), (1 | 2, "no epilogue, no W DMA"), (1 | 4, "no epilogue, no A loads"),
            (1 | 32, "no epilogue, no split"), (1 | 16, "no epilogue, no LDS reads"), (1 | 8, "no epilogue, no MFMA"),
            (1 | 2 | 4 | 16 | 32, "MFMA only"), (2 | 4 | 8 | 16 | 32, "epilogue only")]
EXTRA = os.environ.get("X3_EXTRA", "").split()

def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for bits, _ in VARIANTS:
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DX3_ABLATE=%d" % bits, *EXTRA,
               os.path.join(ROOT, "scream_amd/csrc/gemm_x3.hip"), "-o", os.path.join(OUT, "x3_%d.so" % bits)]
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
    M = int(os.environ.get("X3_M", 327680))
    V, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [("qkv", 768, 256, ops.EPI_ELU1), ("ffn1", 1024, 256, ops.EPI_RELU), ("ffn2", 256, 1024, ops.EPI_RES_LN), ("merge", 256, 256, ops.EPI_RES_LN)]
    data = {}
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        data[name] = (A, ops.split_planes(W), torch.empty(M, N, device=dev), torch.randn(M, 256, device=dev, generator=g), torch.ones(256, device=dev))
    print("%-28s" % "variant" + "".join("%16s" % s[0] for s in shapes) + "   (ms | fp32-equivalent TFLOP/s)")
    for bits, label in VARIANTS:
        lib = ctypes.CDLL(os.path.join(OUT, "x3_%d.so" % bits))
        fn = lib.scream_gemm_x3_f32
        fn.restype = ctypes.c_int
        fn.argtypes = [V, I64, V, V, I64, I64, I32, I32, I32, I32, V, V, I64, V, V, V]
        line = "%-28s" % label
        for name, N, K, epi in shapes:
            A, Wp, o, res, gam = data[name]
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: fn(A.data_ptr(), K, Wp.data_ptr(), o.data_ptr(), N, M, N, K, epi, 512 if epi == ops.EPI_ELU1 else 0, None,
                              res.data_ptr(), 256, gam.data_ptr(), gam.data_ptr(), st)
            assert call() == 0
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): call()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
            t = sorted(ts)[2]
            line += "  %6.3f | %5.1f" % (t, 2.0 * M * N * K / t / 1e9)
        print(line, flush=True)

if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
