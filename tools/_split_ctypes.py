"""ctypes bindings of the split GEMM's entry points in an arbitrary build of gemm_split.hip (the ablation / stamp libraries of
tools/x3_*.py): pack(W) -> image, gemm(...) -> rc, with X3_SPLIT=h2 (default) | x3 choosing the operand split."""
import ctypes, os

V, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
SPLIT = {"h2": 2, "x3": 3}[os.environ.get("X3_SPLIT", "h2")]
A_EXP = 12  # the tools feed standard-normal activations clamped to |a| <= 8
SRC = "scream_amd/csrc/gemm_split.hip"
KERNEL = "17gemm_split_kernelINS_" + {2: "7SplitH2", 3: "8SplitBf3"}[SPLIT]  # mangled-name prefix of the instances that will run


def bind(path):
    import torch
    lib = ctypes.CDLL(path)
    fn = lib.scream_gemm_split_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [V, I64, V, V, I64, I64, I32, I32, I32, I32, V, V, I64, V, V, I32, I32, I32, I32, V]
    pk = lib.scream_pack_w_split
    pk.restype = ctypes.c_int
    pk.argtypes = [V, I32, I32, I32, I32, V, V]

    def pack(W, st):  # every build packs with its own packer (the image layout belongs to the kernel)
        N, K = W.shape
        w_exp = int(torch.floor(torch.log2(32768.0 / W.abs().max())).item()) if SPLIT == 2 else 0
        Wp = torch.empty(2 * SPLIT * N * K, device=W.device, dtype=torch.uint8)
        assert pk(W.data_ptr(), N, K, SPLIT, w_exp, Wp.data_ptr(), st) == 0
        return Wp, w_exp

    def gemm(A, Wp, w_exp, o, M, N, K, epi, n_act, rsd, gam, st):
        return fn(A.data_ptr(), K, Wp.data_ptr(), o.data_ptr(), N, M, N, K, epi, n_act, None, rsd.data_ptr(), 256, gam.data_ptr(),
                  gam.data_ptr(), 0, SPLIT, A_EXP, w_exp, st)
    return lib, pack, gemm
