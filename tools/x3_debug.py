import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import ops
dev = "cuda:0"
for (M, N, K) in [(128, 256, 256), (256, 256, 256), (384, 768, 256), (256, 256, 1024), (1024, 256, 64)]:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5
    ref = A.double() @ W.double().t()
    for rep in range(3):
        out = ops.gemm_x3(A.to(dev), ops.split_planes(W.to(dev))).cpu().double()
        err = (out - ref).abs()
        blk = err.reshape(M // 32, 32, N // 256, 256).amax(dim=(1, 3))
        print(M, N, K, "rep", rep, "max err %.3e" % err.max().item(), "bad 32-row blocks x n-tiles:", (blk > 1e-4).nonzero().tolist()[:24])
