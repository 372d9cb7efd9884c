"""Drop-in for the reference's ``evaluate_3d_match.py`` entry points (lines 31-50, 53-183).

    python evaluate_3d_match.py [--split 3DMatch_test|3DLoMatch_test|3DZeroMatch_test] [--params params/point-generator.pth]
                                [--synthetic N] [--batch-pairs 32] [--no-icp]

Like the reference (evaluate_3d_match.py:106-119) every pose is refined by point-to-point ICP (here on the GPU:
scream_icp_p2p, 0.1 m, 30 iterations) and the refinement is kept only where it improves both RE and TE; RR / RRE / RTE
are those of the refined pose.  ``--no-icp`` (``icp=None`` in the Python API) reports the pre-ICP pose instead.

With the reference's on-disk splits present (``<split>/src%d.npy`` ..., process_3d_match.py:38-40) it evaluates them;
``--synthetic N`` evaluates N seeded synthetic pairs instead (no dataset ships with either repository).
Launch under ``python -m torch.distributed.run --nproc-per-node G`` to shard the pairs over G MI355X.
"""
import argparse
import os

import torch

from scream_amd import dist as _dist
from scream_amd.data import PairFileDataset, SyntheticPairs
from scream_amd.evaluate import RMSE, evaluate_loader  # noqa: F401  (same names as the reference module)
from scream_amd.evaluate import evaluate_3d_lo_match as _lo
from scream_amd.evaluate import evaluate_3d_match as _match
from scream_amd.evaluate import evaluate_3d_zero_match as _zero

_SPLITS = {"3DMatch_test": "3dmatch", "3DLoMatch_test": "lo", "3DZeroMatch_test": "zero"}


def _dataset(split, synthetic=0):
    if synthetic:
        return SyntheticPairs(_SPLITS[split], synthetic)
    if not os.path.exists(os.path.join(split, "info", "scene_names.txt")):
        raise FileNotFoundError("%s/ is not populated (the reference README points at an external download); "
                                "pass --synthetic N to run on seeded synthetic pairs" % split)
    return PairFileDataset(split)


def evaluate_3d_match(net, dis_thresh=0.1, dataset=None, **kw):
    """evaluate_3d_match.py:178-179."""
    return _match(net, dataset or _dataset("3DMatch_test"), dis_thresh, **kw)


def evaluate_3d_lo_match(net, dis_thresh=0.1, dataset=None, **kw):
    """evaluate_3d_match.py:174-175."""
    return _lo(net, dataset or _dataset("3DLoMatch_test"), dis_thresh, **kw)


def evaluate_3d_zero_match(net, dis_thresh=0.1, dataset=None, **kw):
    """evaluate_3d_match.py:182-183."""
    return _zero(net, dataset or _dataset("3DZeroMatch_test"), dis_thresh, **kw)


if __name__ == "__main__":
    from models.pointnet import PointTransformer

    ap = argparse.ArgumentParser()
    ap.add_argument("--split", default="3DMatch_test", choices=sorted(_SPLITS))
    ap.add_argument("--params", default="./params/point-generator.pth")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--batch-pairs", type=int, default=32)
    ap.add_argument("--dis-thresh", type=float, default=0.1)
    ap.add_argument("--no-icp", action="store_true", help="skip the ICP refinement of evaluate_3d_match.py:106-119")
    args = ap.parse_args()
    rank, world, local = _dist.init_from_env()
    device = torch.device("cuda", local)
    net = PointTransformer(d_model=256)
    net.to(device)
    if os.path.exists(args.params):
        net.load_state_dict(torch.load(args.params, map_location=device))
    elif rank == 0:
        print("warning: %s not found, evaluating seeded random weights" % args.params)
    net.eval()
    ds = _dataset(args.split, args.synthetic)
    fn = {"3DMatch_test": evaluate_3d_match, "3DLoMatch_test": evaluate_3d_lo_match, "3DZeroMatch_test": evaluate_3d_zero_match}[args.split]
    fn(net, args.dis_thresh, ds, batch_pairs=args.batch_pairs, icp=None if args.no_icp else "gpu")
