"""Drop-in for the reference's ``evaluate_kitti.py`` (lines 23-110).

    python evaluate_kitti.py [--root KITTI_test] [--params params/kitti-generator.pth] [--synthetic N] [--no-icp]
"""
import argparse
import os

import torch

from scream_amd import dist as _dist
from scream_amd.evaluate_kitti import KittiPairFiles, SyntheticKittiPairs, evaluate  # noqa: F401


def evaluate_test(net, root="KITTI_test", **kw):
    """evaluate_kitti.py:105-110: dis_thresh 1.5, ICP radius 1."""
    return evaluate(net, KittiPairFiles(root, 554), dis_thresh=1.5, icp_thresh=1, **kw)


if __name__ == "__main__":
    from models.pointnet import PointTransformer

    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default="KITTI_test")
    ap.add_argument("--params", default="params/kitti-generator.pth")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--no-icp", action="store_true")
    args = ap.parse_args()
    rank, world, local = _dist.init_from_env()
    device = torch.device("cuda", local)
    net = PointTransformer(d_model=256, self_layer_num=6, cross_layer_num=6)
    net.to(device)
    if os.path.exists(args.params):
        net.load_state_dict(torch.load(args.params, map_location=device))
    elif rank == 0:
        print("warning: %s not found, evaluating seeded random weights" % args.params)
    net.eval()
    ds = SyntheticKittiPairs(args.synthetic) if args.synthetic else KittiPairFiles(args.root, 554)
    evaluate(net, ds, icp=None if args.no_icp else "gpu")
