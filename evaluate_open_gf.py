"""Drop-in for the reference's ``evaluate_open_gf.py`` (lines 46-77).

    python evaluate_open_gf.py [--root OpenGF_test] [--params params/dem-generator.pth] [--synthetic N]
"""
import argparse

import torch

from scream_amd import dist as _dist
from scream_amd.evaluate_open_gf import OpenGFFiles, SyntheticDEM, evaluate_dem_generation  # noqa: F401


def evaluate_DEM_generation(net, root="OpenGF_test", **kw):
    """evaluate_open_gf.py:46: the 650 test patches."""
    return evaluate_dem_generation(net, OpenGFFiles(root, 650), **kw)


if __name__ == "__main__":
    from models.pointnet import DEMTransformer

    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default="OpenGF_test")
    ap.add_argument("--params", default="params/dem-generator.pth")
    ap.add_argument("--synthetic", type=int, default=0)
    args = ap.parse_args()
    rank, world, local = _dist.init_from_env()
    device = torch.device("cuda", local)
    net = DEMTransformer(d_model=256)
    if args.synthetic:
        from scream_amd.synthetic import make_state_dict
        net.load_state_dict(make_state_dict(0, 256, 6, 6, dem=True))
    else:
        net.load_state_dict(torch.load(args.params, map_location="cpu"))
    net = net.to(device).eval()
    ds = SyntheticDEM(args.synthetic) if args.synthetic else OpenGFFiles(args.root, 650)
    evaluate_dem_generation(net, ds)
