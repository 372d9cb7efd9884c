"""Ground-generation evaluation on OpenGF (the reference's ``evaluate_open_gf.py``:46-73) for ``DEMTransformer``.

Per test sample (dsm [N,3], dem_coarse [M,3], dem [N,3], all divided by 50, ``datasets/open_gf.py``:8,66): the model
turns the DSM into a DEM prediction; the metrics are the symmetric squared-distance Chamfer term against the true DEM
and the mean absolute / squared height error of the row-paired points, each x 1000 and averaged over the set.
Here B samples share every kernel launch (the reference runs one per forward), the Chamfer term is two fused 1-NN
searches instead of a dense N x M matrix, and samples are sharded over ranks like registration pairs.

File format (``datasets/open_gf.py``:54-69): ``<root>/<i>.npy`` float [N,6] = (dsm xyz | dem xyz), i = 1..count, and
``<root>/centers/<i>.npy``; dem_coarse is the DEM voxel-downsampled at 20 m -- open3d's ``voxel_down_sample`` in the
reference, restated here (one centroid per occupied voxel, grid origin at min_bound - voxel/2); open3d is not
installed, so that restatement is not pinned against it (point order and grid origin only change which coarse points
the cross-attention sees, not the metric definitions).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import dist as sdist
from .geometry import chamfer_distance, processbar

SCALE_FACTOR = 50.0          # datasets/open_gf.py:8
DEM_COARSE_RESOLUTION = 20   # datasets/open_gf.py:52
METRIC_SCALE = 1000          # evaluate_open_gf.py:50


def voxel_down_sample(points: np.ndarray, voxel: float) -> np.ndarray:
    """Centroid of the points in every occupied voxel; voxel (i,j,k) = floor((p - (min_bound - voxel/2)) / voxel)."""
    pts = np.asarray(points, dtype=np.float64)
    if pts.shape[0] == 0:
        return pts.reshape(0, 3)
    origin = pts.min(axis=0) - voxel * 0.5
    key = np.floor((pts - origin) / voxel).astype(np.int64)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    out = np.zeros((cnt.shape[0], 3))
    np.add.at(out, inv, pts)
    return out / cnt[:, None]


def make_sample(dsm_dem: np.ndarray, center=None):
    """datasets/open_gf.py:58-69 for one [N,6] array."""
    dsm, dem = dsm_dem[:, :3], dsm_dem[:, 3:]
    dem_coarse = voxel_down_sample(dem, DEM_COARSE_RESOLUTION)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a / SCALE_FACTOR, dtype=np.float32))
    return f(dsm), f(dem_coarse), f(dem), center


class OpenGFFiles(torch.utils.data.Dataset):
    def __init__(self, root: str = "OpenGF_test", count: int = 650):
        self.root, self.count = root, count

    def __len__(self):
        return self.count

    def __getitem__(self, item):
        import os
        arr = np.load(os.path.join(self.root, "%d.npy" % (item + 1)))
        cpath = os.path.join(self.root, "centers", "%d.npy" % (item + 1))
        return make_sample(arr, np.load(cpath) if os.path.exists(cpath) else None)


class SyntheticDEM(torch.utils.data.Dataset):
    """Seeded terrain patches (500 m square, smooth relief) with box-shaped 'buildings' and noisy 'vegetation' lifted
    above the ground in the DSM; DEM rows are the ground under the same (x, y)."""

    def __init__(self, n: int, seed0: int = 0, points: int = 4000):
        self.n, self.seed0, self.points = n, seed0, points

    def __len__(self):
        return self.n

    def __getitem__(self, item):
        rng = np.random.default_rng(self.seed0 + item)
        xy = rng.uniform(0, 500, size=(self.points, 2))
        a, b = rng.uniform(0.005, 0.02, size=2)
        ground = 8 * np.sin(a * xy[:, 0] + rng.uniform(0, 6)) + 6 * np.cos(b * xy[:, 1] + rng.uniform(0, 6)) + 0.01 * xy[:, 0]
        lift = np.zeros(self.points)
        for _ in range(6):
            c, half, h = rng.uniform(50, 450, size=2), rng.uniform(10, 40, size=2), rng.uniform(4, 25)
            lift = np.where((np.abs(xy - c) < half).all(axis=1), h, lift)
        veg = rng.random(self.points) < 0.15
        lift = np.where(veg & (lift == 0), rng.uniform(1, 12, size=self.points), lift)
        dsm = np.concatenate([xy, (ground + lift)[:, None]], axis=1)
        dem = np.concatenate([xy, ground[:, None]], axis=1)
        return make_sample(np.concatenate([dsm, dem], axis=1), np.array([250.0, 250.0, 0.0]))


@torch.no_grad()
def evaluate_samples(net, samples: Sequence[tuple], device: Optional[torch.device] = None) -> np.ndarray:
    """One batch of samples -> [B, 3] (chamfer, height MAE, height MSE), each x 1000 (evaluate_open_gf.py:57-68)."""
    device = device or next(net.parameters()).device
    dsms = [s[0].to(device) for s in samples]
    coarse = [s[1].to(device) for s in samples]
    dems = [s[2].to(device) for s in samples]
    preds = net.forward_batch(dsms, coarse)
    rows = torch.zeros(len(samples), 3, dtype=torch.float64)
    for i, (p, d) in enumerate(zip(preds, dems)):
        dz = p[:, 2] - d[:, 2]
        rows[i, 0] = chamfer_distance(p[None], d[None]).double().cpu()
        rows[i, 1] = dz.abs().mean().double().cpu()
        rows[i, 2] = (dz * dz).mean().double().cpu()
    return rows.numpy() * METRIC_SCALE


def evaluate_dem_generation(net, dataset, batch_samples: int = 8, verbose: bool = True):
    """evaluate_open_gf.py:46-73 -> (chamfer_loss, high_loss_mae, high_loss_mse).  With torch.distributed initialised the
    samples are sharded round-robin over ranks and the per-sample rows are all-gathered once at the end."""
    n = len(dataset)
    rank, world = sdist.rank_world()
    mine = sdist.shard_indices(n, rank, world)
    rows: List[np.ndarray] = []
    for lo in range(0, len(mine), batch_samples):
        ids = mine[lo:lo + batch_samples]
        r = evaluate_samples(net, [dataset[i] for i in ids])
        full = np.zeros((len(ids), sdist.ROW_WIDTH))
        full[:, sdist.COL_PAIR] = ids
        full[:, 1:4] = r
        rows.append(full)
        if verbose and rank == 0:
            acc = np.concatenate(rows)[:, 1:4].mean(axis=0)
            print("\r%s  chamfer loss: %.5f  high_loss_mae: %.5f  high_loss_mse: %.5f" % (
                processbar(min(lo + batch_samples, len(mine)), len(mine)), acc[0], acc[1], acc[2]), end="")
    local = np.concatenate(rows) if rows else np.zeros((0, sdist.ROW_WIDTH))
    allrows = sdist.all_gather_rows(local)
    out = tuple(float(v) for v in allrows[:, 1:4].sum(axis=0) / max(n, 1))
    if verbose and rank == 0:
        print("\ntest finished ! chamfer loss: %.5f  high loss mae: %.5f  hige_loss_mse: %.5f" % out)
    return out
