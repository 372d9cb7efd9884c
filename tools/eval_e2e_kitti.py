#!/usr/bin/env python3
"""End-to-end evaluate_kitti.evaluate throughput (the KITTI harness of evaluate_kitti.py:23-110 on the batched path): synthetic
KITTI-like pairs (13-16 k points per cloud, BASELINE configs[3]) held in memory, everything included -- bbox normalisation done,
host packing, forward, 1-NN at 1.5, Kabsch, GPU ICP with the reference's 1 m radius and 1000-iteration cap (or icp=None), RE/TE,
success bookkeeping.  usage: eval_e2e_kitti.py [n_pairs] [--distinct D] [--batch B] [--gen-procs P]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gen(i):
    from scream_amd.evaluate_kitti import SyntheticKittiPairs
    return SyntheticKittiPairs(i + 1)[i]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=512)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--gen-procs", type=int, default=14)
    args = ap.parse_args()
    D = min(args.distinct, args.n)
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(args.gen_procs, D)) as pool:  # before anything touches the GPU
        items = pool.map(gen, range(D))
    import numpy as np, torch
    from scream_amd import evaluate_kitti as ek
    from scream_amd.model import PointTransformer
    from scream_amd.synthetic import make_state_dict

    class Mem(torch.utils.data.Dataset):
        def __len__(self): return args.n
        def __getitem__(self, i): return items[i % D]
    net = PointTransformer(256, 6, 6); net.load_state_dict(make_state_dict(0, 256, 6, 6)); net = net.to("cuda:0").eval()
    print("evaluate_kitti.evaluate end to end, %d pairs per leg (%d distinct synthetic KITTI-like pairs, mean %.0f + %.0f points), batch %d, gemm_backend %s"
          % (args.n, D, np.mean([it[0].shape[0] for it in items]), np.mean([it[1].shape[0] for it in items]), args.batch, net.gemm_backend), flush=True)
    ek.evaluate(net, torch.utils.data.Subset(Mem(), range(2 * args.batch)), batch_pairs=args.batch, skip=(), verbose=False)  # warm-up
    for tag, kw in (("icp=None", dict(icp=None)), ("icp='gpu' (reference default: 1 m, <= 1000 iterations)", dict(icp="gpu")),
                    ("icp='gpu', autocast=True (the reference's fp16 autocast mirror; labelled, not fp32-accurate)", dict(icp="gpu", autocast=True))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ek.evaluate(net, Mem(), batch_pairs=args.batch, skip=(), verbose=False, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-100s %5d pairs in %6.2f s -> %7.1f pairs/s   (loss, rre, rte, success) = %s"
              % (tag, args.n, dt, args.n / dt, tuple(round(float(v), 4) for v in out)), flush=True)


if __name__ == "__main__":
    main()
