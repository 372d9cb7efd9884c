#!/usr/bin/env python3
"""Normwise error against float64 of the operand-split products with the hardware's own MFMA accumulation:
3 x bf16 / 6 products (the round-2 kernels) vs 2 x fp16 / 3 products with exact power-of-two operand scales (round 3),
on the data of tests/test_gpu_configs.py::test_split_gemm_is_fp32_accurate_against_float64 (rows of mixed scale).
`build` here, run on the GPU.  Also probes whether the fp16 MFMA and the f32 -> f16 conversion keep subnormals."""
import ctypes, math, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "split_acc.so")


def build():
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           os.path.join(HERE, "split_acc.hip"), "-o", SO])


def pow2_scale(t, top=32768.0):
    return 2.0 ** math.floor(math.log2(top / float(t.abs().max())))


def run():
    import torch
    dev = "cuda:0"
    lib = ctypes.CDLL(SO)
    V = ctypes.c_void_p
    lib.split_gemm_launch.argtypes = [ctypes.c_int, V, V, V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, V]
    lib.denorm_probe_launch.argtypes = [V, V]
    st = torch.cuda.current_stream().cuda_stream
    o = torch.zeros(4, device=dev)
    o[1] = 2.0 ** -20
    lib.denorm_probe_launch(o.data_ptr(), st)
    torch.cuda.synchronize()
    print("fp16 subnormal input to v_mfma_f32_32x32x16_f16: 16 x (2^-20 x 2^14) = %g (0.25 if kept, 0 if flushed); "
          "(half)(2^-20) -> %g (9.5367e-07 if kept)" % (o[0].item(), o[2].item()))
    names = {0: "bf16x3 / 6 products", 1: "fp16x2 / 3 products", 2: "fp16x2 / 4 products", 3: "fp16x1"}
    print("%-28s %-22s %12s %12s" % ("data", "split", "max err/scale", "rms err/scale"))
    for (N, K) in [(768, 256), (256, 256), (512, 256), (1024, 256), (256, 1024)]:
        g = torch.Generator().manual_seed(1000 * N + K)
        M = 4096
        A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-6, 7, (M, 1), generator=g).float())
        W = torch.randn(N, K, generator=g) / K ** 0.5
        C64 = A.double() @ W.double().t()
        scale = A.double().abs() @ W.double().abs().t()
        Ad, Wd = A.to(dev), W.to(dev)
        C = torch.empty(M, N, device=dev)
        sa_auto, sw_auto = pow2_scale(A), pow2_scale(W)
        for mode, sa, sw, tag in [(0, 1.0, 1.0, ""), (1, sa_auto, sw_auto, "scales 2^%d, 2^%d" % (math.log2(sa_auto), math.log2(sw_auto))),
                                  (2, sa_auto, sw_auto, "same"), (1, 1.0, 1.0, "unscaled"), (1, sa_auto / 256, sw_auto / 256, "scales / 2^8"),
                                  (3, sa_auto, sw_auto, "")]:
            lib.split_gemm_launch(mode, Ad.data_ptr(), Wd.data_ptr(), C.data_ptr(), M, N, K, sa, sw, st)
            torch.cuda.synchronize()
            e = ((C.cpu().double() - C64).abs() / scale)
            print("%-28s %-22s %12.3e %12.3e  %s" % ("mixed rows N%d K%d" % (N, K), names[mode], e.max().item(), e.pow(2).mean().sqrt().item(), tag), flush=True)
        e32 = (((A @ W.t()).double() - C64).abs() / scale)
        print("%-28s %-22s %12.3e %12.3e" % ("", "torch-CPU fp32", e32.max().item(), e32.pow(2).mean().sqrt().item()))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
