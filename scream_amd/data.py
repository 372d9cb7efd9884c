"""Test-set items in the reference's convention (datasets/three_d_match.py:219-294).

``__getitem__`` of every dataset here yields the reference's 9-tuple
    (src [N,3], tgt [M,3], rot [3,3], trans [3,1], s, idx [2], covariance [6,6], c [3], scene_idx)
with the unit-ball normalisation of three_d_match.py:233-240, so ``evaluate_loader`` consumes real
on-disk splits and the seeded synthetic ones alike.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

from . import synthetic

SCENE_NAMES = ["Kitchen", "Home_1", "Home_2", "Hotel_1", "Hotel_2", "Hotel_3", "Study", "MIT_Lab"]  # evaluate_3d_match.py:62
# benchmark directory name -> scene slot (datasets/three_d_match.py keeps the same table as scene_name_to_idx)
SCENE_DIR_TO_IDX = {
    "7-scenes-redkitchen": 0, "sun3d-home_at-home_at_scan1_2013_jan_1": 1, "sun3d-home_md-home_md_scan9_2012_sep_30": 2,
    "sun3d-hotel_uc-scan3": 3, "sun3d-hotel_umd-maryland_hotel1": 4, "sun3d-hotel_umd-maryland_hotel3": 5,
    "sun3d-mit_76_studyroom-76-1studyroom2": 6, "sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika": 7,
}


def normalize_pair(src: np.ndarray, tgt: np.ndarray, T: np.ndarray, mode: str = "ball"):
    """Centre/scale a pair and move the GT translation with it; fp64 numpy in, fp32 tensors out.
    mode "ball" (datasets/three_d_match.py:231-240): c = mean of (registered src U tgt), s = 1 / max radius.
    mode "bbox" (datasets/kitti.py:268-273,340-346 ``norm_pc``): c = bounding-box centre, s = 2 / largest extent.
    Returns (src, tgt, rot, trans, s, c)."""
    rot, trans = T[:3, :3], T[:3, 3:]
    merged = np.concatenate([(rot @ src.T + trans).T, tgt], axis=0)
    if mode == "ball":
        c = merged.mean(axis=0)
        s = 1.0 / float(np.linalg.norm(merged - c[None], axis=1).max())
    elif mode == "bbox":
        hi, lo = merged.max(axis=0), merged.min(axis=0)
        c = (lo + hi) / 2
        s = 1.0 / (float((hi - lo).max()) / 2)
    else:
        raise ValueError(mode)
    src_n, tgt_n = s * (src - c), s * (tgt - c)
    trans_n = s * (trans - c.reshape(3, 1) + rot @ c.reshape(3, 1))
    return (torch.Tensor(src_n), torch.Tensor(tgt_n), torch.Tensor(rot), torch.Tensor(trans_n), s, torch.Tensor(c))


class PairFileDataset(Dataset):
    """A split directory in the reference's on-disk format (process_3d_match.py:38-40,199-200):
    src%d.npy / tgt%d.npy float64 [N,3], T%d.npy [4,4], info/idx%d.npy, info/covariance%d.npy,
    info/scene_names.txt (one benchmark scene directory name per pair)."""

    def __init__(self, root: str):
        self.root = root
        with open(os.path.join(root, "info", "scene_names.txt")) as f:
            self.scene_names = [l.strip() for l in f if l.strip()]

    def __len__(self):
        return len(self.scene_names)

    def __getitem__(self, i):
        r = self.root
        src, tgt, T = (np.load(os.path.join(r, "%s%d.npy" % (k, i))) for k in ("src", "tgt", "T"))
        idx = np.load(os.path.join(r, "info", "idx%d.npy" % i))
        cov = np.load(os.path.join(r, "info", "covariance%d.npy" % i))
        src_n, tgt_n, rot, trans, s, c = normalize_pair(src, tgt, T)
        return (src_n, tgt_n, rot, trans, s, torch.LongTensor(idx), torch.Tensor(cov), c,
                SCENE_DIR_TO_IDX[self.scene_names[i]])


class SyntheticPairs(Dataset):
    """Seeded stand-in for 3DMatch_test / 3DLoMatch_test / 3DZeroMatch_test (kind = "3dmatch" | "lo" | "zero"):
    pair i is generated from seed0 + i, so any rank can materialise any shard without I/O."""

    def __init__(self, kind: str = "3dmatch", count: int = 16, seed0: int = 0):
        self.kind, self.count, self.seed0 = kind, count, seed0

    def __len__(self):
        return self.count

    def __getitem__(self, i):
        src, tgt, T, idx, cov, scene = synthetic.make_3dmatch_pair(self.seed0 + i, self.kind)
        src_n, tgt_n, rot, trans, s, c = normalize_pair(src, tgt, T)
        return (src_n, tgt_n, rot, trans, s, torch.LongTensor(idx), torch.Tensor(cov), c, scene)


_PACKED = "scream-packed-batch-v1"


def collate_pairs(items: Sequence[tuple]):
    """Var-len pairs cannot be stacked: in the main process a batch is simply the list of 9-tuples.  In a DataLoader WORKER
    the batch is flattened into three tensors (every float of every item back to back, the integer metadata, the scales):
    a list of 32 nine-tuples is 288 tensors, each its own shared-memory segment and file descriptor on the way to the main
    process and again through the pinning thread -- measured 600 pairs/s from files with workers against 1 300 without;
    three tensors per batch cost nothing.  ``unpack_batch`` turns them back into the same 9-tuples (views of the one
    buffer, bit-identical values)."""
    if torch.utils.data.get_worker_info() is None:
        return list(items)
    flats, meta, scales = [], [], []
    for src, tgt, rot, trans, s, idx, cov, c, scene in items:
        flats += [src.reshape(-1), tgt.reshape(-1), rot.reshape(-1), trans.reshape(-1), c.reshape(-1), cov.reshape(-1)]
        meta.append([src.shape[0], tgt.shape[0], int(idx[0]), int(idx[1]), int(scene), cov.shape[0], cov.shape[1]])
        scales.append(float(s))
    return (_PACKED, torch.cat([f.float() for f in flats]), torch.tensor(meta, dtype=torch.int64), torch.tensor(scales, dtype=torch.float64))


def unpack_batch(batch) -> List[tuple]:
    """Inverse of the worker-side collation (a plain list of 9-tuples passes through)."""
    if not (isinstance(batch, (tuple, list)) and len(batch) == 4 and isinstance(batch[0], str) and batch[0] == _PACKED):
        return list(batch)
    _, flat, meta, scales = batch
    out, o = [], 0

    def take(n, *shape):
        nonlocal o
        v = flat[o:o + n].view(*shape)
        o += n
        return v
    for (n, m, i0, i1, scene, cr, cc), s in zip(meta.tolist(), scales.tolist()):
        src, tgt = take(3 * n, n, 3), take(3 * m, m, 3)
        rot, trans, c, cov = take(9, 3, 3), take(3, 3, 1), take(3, 3), take(cr * cc, cr, cc)
        out.append((src, tgt, rot, trans, s, torch.LongTensor([i0, i1]), cov, c, scene))
    return out
