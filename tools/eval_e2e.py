#!/usr/bin/env python3
"""End-to-end evaluate_loader throughput on pre-generated synthetic 3DMatch-like items (host work included:
H2D copies, packing, metric rows, RMSE) -- compare with bench.py's device-resident pairs/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd.data import SyntheticPairs
from scream_amd.evaluate import evaluate_loader
from scream_amd.model import PointTransformer
from scream_amd.synthetic import make_state_dict
import multiprocessing as mp

def gen(i):
    return SyntheticPairs("3dmatch", 1, seed0=i)[0]

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    with mp.get_context("spawn").Pool(14) as pool:
        items = pool.map(gen, range(n))
    class Mem(torch.utils.data.Dataset):
        def __len__(self): return len(items)
        def __getitem__(self, i): return items[i]
    net = PointTransformer(256, 6, 6); net.load_state_dict(make_state_dict(0, 256, 6, 6)); net = net.to("cuda:0").eval()
    evaluate_loader(net, Mem(), batch_pairs=32, verbose=False)  # warm-up
    for icp in (None, "gpu"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = evaluate_loader(net, Mem(), batch_pairs=32, verbose=False, icp=icp)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("evaluate_loader icp=%s: %d pairs in %.2f s -> %.1f pairs/s   (loss, rre, rte, rr) = %s" % (icp, n, dt, n / dt, tuple(round(float(v), 4) for v in out)))
