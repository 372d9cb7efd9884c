#!/usr/bin/env python3
"""End-to-end evaluate_loader throughput through the Python boundary -- host packing, H2D copies, A1-A10 (+ GPU ICP), metric
rows, RMSE, aggregation all included -- to set beside bench.py's device-resident pairs/s.  Three legs, each >= 2 048 pairs:

  1. in-memory items, icp=None          (pre-ICP metrics)
  2. in-memory items, icp="gpu"         (the reference's default: every pose refined, evaluate_3d_match.py:106-119)
  3. FROM FILES: a split directory written in the reference's on-disk format (process_3d_match.py:38-40,199-200; float64
     src%d.npy / tgt%d.npy, T%d.npy, info/idx%d.npy, info/covariance%d.npy, info/scene_names.txt) read back through
     PairFileDataset + evaluate_loader(num_workers=k) (np.load + the per-item normalisation of
     datasets/three_d_match.py:228-242 in worker processes, pinned batches), for several k: which k sustains >= 95 % of leg 2.

The synthetic pairs are `--distinct` (default 256) seeded 3DMatch-like pairs, each visited n / distinct times (the files stay in
the page cache, as a test split of 1 253 small pairs does on any host).  usage: eval_e2e.py [n_pairs] [--distinct D] [--gen-procs P]"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def gen_raw(i):
    """Raw (un-normalised, float64) pair + metadata, as process_3d_match.py would write it."""
    from scream_amd import synthetic
    return synthetic.make_3dmatch_pair(i, "3dmatch")


class CycledFiles:  # (module level: picklable for spawned loader workers) every file of the split directory n / D times
    def __init__(self, root, n, d):
        self.root, self.n, self.d, self.files = root, n, d, None

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if self.files is None:
            from scream_amd.data import PairFileDataset
            self.files = PairFileDataset(self.root)
        return self.files[i % self.d]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=2048)
    ap.add_argument("--distinct", type=int, default=256)
    ap.add_argument("--gen-procs", type=int, default=14, help="1 = generate in-process (e.g. under rocprofv3)")
    ap.add_argument("--workers", default="0,2,4,8,12")
    ap.add_argument("--torch-threads", type=int, default=0, help="torch.set_num_threads() of the main process (0 = leave)")
    ap.add_argument("--in-flight", default="4", help="batches kept enqueued (comma list: legs 1-2 are run for each)")
    ap.add_argument("--worker-context", default=None, help="multiprocessing context of the loader workers (fork | spawn | forkserver)")
    args = ap.parse_args()
    D = min(args.distinct, args.n)
    if args.gen_procs > 1:  # before anything touches the GPU
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(args.gen_procs) as pool:
            raw = pool.map(gen_raw, range(D))
    else:
        raw = [gen_raw(i) for i in range(D)]

    import torch
    from scream_amd.data import SCENE_DIR_TO_IDX, PairFileDataset, normalize_pair
    from scream_amd.evaluate import evaluate_loader
    from scream_amd.model import PointTransformer
    from scream_amd.synthetic import make_state_dict
    scene_dirs = {v: k for k, v in SCENE_DIR_TO_IDX.items()}
    if args.torch_threads:
        torch.set_num_threads(args.torch_threads)
    print("host: %d cores in the affinity mask, torch intra-op threads %d, OMP_NUM_THREADS=%s" % (len(os.sched_getaffinity(0)), torch.get_num_threads(), os.environ.get("OMP_NUM_THREADS")), flush=True)

    def item(r):
        src, tgt, T, idx, cov, scene = r
        s_n, t_n, rot, trans, s, c = normalize_pair(src, tgt, T)
        return (s_n, t_n, rot, trans, s, torch.LongTensor(idx), torch.Tensor(cov), c, scene)
    items = [item(r) for r in raw]

    class Mem(torch.utils.data.Dataset):
        def __len__(self): return args.n
        def __getitem__(self, i): return items[i % D]

    net = PointTransformer(256, 6, 6); net.load_state_dict(make_state_dict(0, 256, 6, 6)); net = net.to("cuda:0").eval()
    print("evaluate_loader end to end, %d pairs per leg (%d distinct synthetic 3DMatch-like pairs, mean %.0f + %.0f points), batch 32, gemm_backend %s"
          % (args.n, D, np.mean([r[0].shape[0] for r in raw]), np.mean([r[1].shape[0] for r in raw]), net.gemm_backend), flush=True)
    evaluate_loader(net, torch.utils.data.Subset(Mem(), range(64)), batch_pairs=32, verbose=False)  # warm-up

    def leg(tag, ds, **kw):
        # `steady`: from the moment the first batch reaches the GPU path (worker processes forked, first files read) to the end --
        # starting a DataLoader's worker processes costs 1-2 s once per pass, whatever the dataset's length
        first = []

        def hook(batch, src_pred, pair_ids):
            if not first:
                first.append((time.perf_counter(), len(pair_ids)))
            return src_pred
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = evaluate_loader(net, ds, batch_pairs=32, verbose=False, pred_hook=hook, **kw)
        torch.cuda.synchronize(); t1 = time.perf_counter(); dt = t1 - t0
        steady = (len(ds) - 32) / (t1 - first[0][0])
        print("%-46s %5d pairs in %6.2f s -> %7.1f pairs/s  (steady %7.1f, start-up %.2f s)   (loss, rre, rte, rr) = %s"
              % (tag, len(ds), dt, len(ds) / dt, steady, first[0][0] - t0, tuple(round(float(v), 4) for v in out)), flush=True)
        return steady, out
    flights = [int(v) for v in args.in_flight.split(",")]
    for nf in flights[::-1]:  # (the first of the list last: it is the one the from-files legs are compared with)
        leg("1. in-memory, icp=None, in_flight=%d" % nf, Mem(), icp=None, in_flight=nf)
        mem_rate, mem_out = leg("2. in-memory, icp='gpu' (reference default), in_flight=%d" % nf, Mem(), icp="gpu", in_flight=nf)

    with tempfile.TemporaryDirectory() as root:  # the reference's on-disk layout, process_3d_match.py:38-40,199-200
        os.makedirs(os.path.join(root, "info"))
        for i, (src, tgt, T, idx, cov, scene) in enumerate(raw):
            np.save(os.path.join(root, "src%d.npy" % i), np.asarray(src, dtype=np.float64))
            np.save(os.path.join(root, "tgt%d.npy" % i), np.asarray(tgt, dtype=np.float64))
            np.save(os.path.join(root, "T%d.npy" % i), np.asarray(T, dtype=np.float64))
            np.save(os.path.join(root, "info", "idx%d.npy" % i), np.asarray(idx))
            np.save(os.path.join(root, "info", "covariance%d.npy" % i), np.asarray(cov))
        with open(os.path.join(root, "info", "scene_names.txt"), "w") as f:
            f.write("\n".join(scene_dirs[r[5]] for r in raw) + "\n")
        best = None
        for k in [int(v) for v in args.workers.split(",")]:
            rate, out = leg("3. from files, icp='gpu', num_workers=%d%s" % (k, " (%s)" % args.worker_context if args.worker_context and k else ""),
                            CycledFiles(root, args.n, D), icp="gpu", num_workers=k, worker_context=args.worker_context, in_flight=flights[0])
            assert all(abs(float(a) - float(b)) < 1e-9 for a, b in zip(out, mem_out)), "from-files results differ from in-memory"
            if best is None and rate >= 0.95 * mem_rate:
                best = k
        print("smallest num_workers whose steady rate is >= 95 %% of the in-memory steady rate (%.1f pairs/s): %s" % (mem_rate, best), flush=True)


if __name__ == "__main__":
    main()
