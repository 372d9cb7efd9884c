"""KITTI evaluation harness (evaluate_kitti.py:23-110) on the batched MI355X path -- SURVEY.md section 8f, row 2.

Differences from the 3DMatch harness, all from the reference: items are 6-tuples (src, tgt, rot, trans, s, c) with
the bounding-box normalisation of datasets/kitti.py:268-273; src_center = -(R^T t)^T (evaluate_kitti.py:39);
dis_thresh 1.5; ICP radius 1 m with up to 1000 iterations (:64-70); success = RE <= 5 deg and TE <= 2 m (:81);
items 124 and 142 are skipped (:32-34).  The reference wraps the forward in fp16 autocast (:37: fp16 matrix products on CUDA, a
no-op on the CPU path parity is defined on); this path stays fp32-accurate by default, and ``autocast=True`` mirrors the
reference's mode (one fp16 plane per operand, fp32 accumulation: gemm_backend "h1", csrc/split.h) for the duration of the
call.  Returns (point_trans_loss, success_rre, success_rte, success_rate),
the numbers the reference prints (:97-102).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

from . import dist as sdist
from . import synthetic
from .data import normalize_pair
from . import lanes as _lanes
from .evaluate import IN_FLIGHT, register_items_async
from .geometry import processbar

SKIP_ITEMS = (124, 142)  # evaluate_kitti.py:32-34
KITTI_DIS_THRESH, KITTI_ICP_DIST, KITTI_ICP_ITERS = 1.5, 1.0, 1000  # evaluate_kitti.py:64-70,106-110


class KittiPairFiles(Dataset):
    """<root>/src%d.npy, tgt%d.npy, T%d.npy as written by process_kitti.py; 554 test pairs (datasets/kitti.py:333)."""

    def __init__(self, root: str, count: Optional[int] = None):
        self.root = root
        self.count = count if count is not None else len([f for f in os.listdir(root) if f.startswith("T") and f.endswith(".npy")])

    def __len__(self):
        return self.count

    def __getitem__(self, i):
        src, tgt, T = (np.load(os.path.join(self.root, "%s%d.npy" % (k, i))) for k in ("src", "tgt", "T"))
        return normalize_pair(src, tgt, T, "bbox")


class SyntheticKittiPairs(Dataset):
    """Seeded LiDAR-like stand-in for KITTI_test (voxel 0.7 m, ~13-16k points per cloud)."""

    def __init__(self, count: int = 8, seed0: int = 0):
        self.count, self.seed0 = count, seed0

    def __len__(self):
        return self.count

    def __getitem__(self, i):
        return normalize_pair(*synthetic.make_kitti_pair(self.seed0 + i), "bbox")


def _strip6(item):
    src, tgt, rot, trans, s, c = item
    if torch.is_tensor(src) and src.dim() == 3:
        src, tgt, rot, trans, c = src[0], tgt[0], rot[0], trans[0], c[0]
        s = s[0] if torch.is_tensor(s) or isinstance(s, (list, tuple)) else s
    return src.float(), tgt.float(), rot.float(), trans.float(), (float(s.item()) if torch.is_tensor(s) else float(s)), c.float()


@torch.no_grad()
def evaluate(net, loader, dis_thresh: float = KITTI_DIS_THRESH, icp_thresh: float = KITTI_ICP_DIST, icp="gpu",
             icp_iters: int = KITTI_ICP_ITERS, batch_pairs: int = 8, skip: Sequence[int] = SKIP_ITEMS,
             verbose: bool = True, pred_hook=None, autocast: bool = False, in_flight: int = IN_FLIGHT):
    """evaluate_kitti.py:23-103.  Pairs are sharded round-robin over ranks when torch.distributed is initialised.
    autocast=True: the forward's matrix products in fp16 with fp32 accumulation, like the reference's `with autocast()`
    (:37) -- a labelled reduced-precision mode, tolerance-tested against the default path, never the default."""
    # the arithmetic is an argument of every forward of this call, never a module attribute: the model caches one weight image per
    # backend (PointTransformer._pack_weights), so the switch repacks nothing, frees nothing that a queued forward may still read,
    # and a net shared with other callers or threads keeps its own mode
    backend = "h1" if autocast else None
    dataset = getattr(loader, "dataset", loader)
    rank, world = sdist.rank_world()
    ids = [i for i in range(len(dataset)) if i not in skip]
    mine = ids[rank::world]
    rows = []

    def collect(fin, bid, n_done):
        T, T_gt, re, te, loss = fin()
        for k, i in enumerate(bid):
            r = np.zeros(sdist.ROW_WIDTH)
            r[sdist.COL_PAIR], r[sdist.COL_SUCCESS] = i, float(re[k] <= 5.0 and te[k] <= 2.0)
            r[sdist.COL_RE], r[sdist.COL_TE], r[sdist.COL_LOSS] = re[k], te[k], loss[k]
            rows.append(r)
        if verbose and rank == 0:
            print("\r%s  re: %.5f  te: %.5f" % (processbar(n_done, len(mine)), re[-1], te[-1]), end="")

    # like evaluate_loader (scream_amd/evaluate.py): in_flight batches enqueued, each whole on its own stream, collected in order
    dev = next(net.parameters()).device
    in_flight = max(1, int(in_flight))
    streams = _lanes.lane_streams(dev, in_flight) if dev.type == "cuda" else [None] * in_flight
    pending = []
    for k, b0 in enumerate(range(0, len(mine), batch_pairs)):
        bid = mine[b0:b0 + batch_pairs]
        its = [_strip6(dataset[i]) for i in bid]
        centers = [-(it[2].t() @ it[3]).reshape(3) for it in its]  # evaluate_kitti.py:39
        fin = register_items_async(net, its, centers, bid, "tgt", dis_thresh, icp, icp_thresh, icp_iters, pred_hook=pred_hook,
                                   stream=streams[k % in_flight], backend=backend)
        pending.append((fin, bid, min(b0 + batch_pairs, len(mine))))
        if len(pending) >= in_flight:
            collect(*pending.pop(0))
    for pd in pending:
        collect(*pd)
    allrows = sdist.all_gather_rows(np.array(rows).reshape(-1, sdist.ROW_WIDTH))
    n = max(allrows.shape[0], 1)
    ok = allrows[:, sdist.COL_SUCCESS] > 0
    n_ok = max(int(ok.sum()), 1)
    out = (float(allrows[:, sdist.COL_LOSS].sum() / n), float(allrows[ok, sdist.COL_RE].sum() / n_ok),
           float(allrows[ok, sdist.COL_TE].sum() / n_ok), float(ok.sum() / n))
    if verbose and rank == 0:
        print("\ntest finish  loss: %.5f  rre: %.5f  rte: %.5f  success rate: %.5f" % out)
    return out
