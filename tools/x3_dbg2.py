import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(134400, 512, 64), (134400, 256, 64), (65536, 512, 64), (134400, 512, 256)]:
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5; Wp = ops.split_planes(W)
    bad3 = badf = 0
    for rep in range(30):
        A = torch.randn(M, K, device=dev)
        ref = A @ W.t()
        a = ops.gemm_x3(A, Wp); b = ops.gemm_f32(A, W)
        e3 = (a - ref).abs().max().item(); ef = (b - ref).abs().max().item()
        if not e3 < 1e-3:
            bad3 += 1
            if bad3 == 1:
                blk = (a - ref).abs().reshape(M // 32, 32, N // 256, 256).amax(dim=(1, 3))
                idx = (blk > 1e-3).nonzero()
                print("   x3 bad 32-row blocks (first 12):", idx[:12].tolist(), "count", idx.shape[0], "tiles(256 rows):", sorted(set((idx[:, 0] // 8).tolist()))[:10])
        if not ef < 1e-3: badf += 1
    print(M, N, K, "x3 bad runs", bad3, "f32 bad runs", badf)
