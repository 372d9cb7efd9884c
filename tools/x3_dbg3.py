import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd import ops
dev = "cuda:0"
which = sys.argv[1]
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(134400, 512, 64), (87040, 256, 256), (153600, 1024, 256), (100096, 256, 1024)]
data = []
for (M, N, K) in shapes:
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    A = torch.randn(M, K, device=dev, generator=g)
    data.append((A, W, ops.split_planes(W), A @ W.t()))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
fn = (lambda A, W, Wp: ops.gemm_x3(A, Wp)) if which == "x3" else (lambda A, W, Wp: ops.gemm_f32(A, W))
t0 = time.time(); bad = 0; n = 0
while time.time() - t0 < 30:
    outs = []
    for si, st in enumerate(streams):
        with torch.cuda.stream(st):
            for j in range(4):
                A, W, Wp, ref = data[(n + si + j) % len(data)]
                outs.append((fn(A, W, Wp), ref, (n + si + j) % len(data)))
    torch.cuda.synchronize()
    for o, ref, k in outs:
        e = (o - ref).abs().max().item()
        if not e < 1e-3:
            bad += 1
            if bad <= 5:
                M, N, K = shapes[k]
                blk = (o - ref).abs().reshape(M // 32, 32, N // 256, 256).amax(dim=(1, 3))
                idx = (blk > 1e-3).nonzero()
                print("BAD", which, shapes[k], "err %.3g" % e, "bad (32-row block, n-tile) count", idx.shape[0], "first", idx[:6].tolist(), "256-row tiles", sorted(set((idx[:, 0] // 8).tolist()))[:8], flush=True)
    n += 1
print(which, "iterations", n, "bad outputs", bad)
