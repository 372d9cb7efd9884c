"""Drop-in equivalents of the reference's geometry helpers (utils.py:72-78,112-189) on the HIP kernels,
plus the fused batched search+solve stage of the evaluation loop (evaluate_3d_match.py:94-104)."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from .packing import PackedBatch

__all__ = ["square_distance", "nn_search_pair", "chamfer_distance", "rigid_transform_3d", "integrate_trans", "transformation_error",
           "register_from_prediction", "register_batch", "processbar"]


def _i32(vals, dev):
    return torch.tensor(list(vals), dtype=torch.int32, device=dev)


def nn_search_pair(query: torch.Tensor, ref: torch.Tensor, s: float, thresh: float):
    """square_distance(query / s, ref / s)[0].min(dim=1) and `< thresh` (evaluate_3d_match.py:94-95)
    for ONE pair, query [N,3], ref [M,3]; returns (dmin [N] fp32, idx [N] int64, valid [N] bool)."""
    dev = query.device
    q = query.reshape(-1, 3).contiguous().float()
    r = ref.reshape(-1, 3).contiguous().float()
    idx, dmin, valid = ops.nn_search(q, r, _i32([0], dev), _i32([q.shape[0]], dev), _i32([0], dev),
                                     _i32([r.shape[0]], dev), torch.tensor([s], dtype=torch.float32, device=dev),
                                     q.shape[0], r.shape[0], thresh)
    return dmin, idx.long(), valid.bool()


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """utils.py:72-78: dense [B,N,M] squared distances (same fp32 rounding sequence as the fused search).
    API compatibility: the evaluation path itself uses the fused ``nn_search`` and never builds this matrix."""
    B, N, _ = src.shape
    _, M, _ = dst.shape
    out = torch.empty(B, N, M, device=src.device, dtype=torch.float32)
    _lib.check(_lib.load().scream_square_distance(ops._p(src.contiguous().float()), ops._p(dst.contiguous().float()),
                                                  ops._p(out), B, N, M, ops._stream()), "scream_square_distance")
    return out


def chamfer_distance(f: torch.Tensor, f_: torch.Tensor) -> torch.Tensor:
    """The symmetric squared-distance Chamfer term of evaluate_open_gf.py:25-41 (``ChamferDistance``):
    mean_n min_m |f_n - f'_m|^2 + mean_m min_n |f'_m - f_n|^2 for ONE pair of clouds [1,N,3], [1,M,3] -- two fused
    1-NN searches instead of the reference's dense N x M matrix and its two ``min``."""
    d_ab, _, _ = nn_search_pair(f[0], f_[0], 1.0, float("inf"))
    d_ba, _, _ = nn_search_pair(f_[0], f[0], 1.0, float("inf"))
    return d_ab.mean() + d_ba.mean()


def integrate_trans(R, t):
    """utils.py:112-135: [R | t; 0 0 0 1] for one pose ([3,3], [3,1]) or a batch ([bs,3,3], [bs,3,1]); torch tensors
    give a float32 tensor on R's device (the reference fills a torch.eye), numpy arrays a float64 array."""
    is_torch = isinstance(R, torch.Tensor)
    batch_shape = tuple(R.shape[:-2])
    if is_torch:
        out = torch.zeros(batch_shape + (4, 4), dtype=torch.float32, device=R.device)
        t = t.reshape(batch_shape + (3,)).to(out.dtype)
    else:
        out = np.zeros(batch_shape + (4, 4))
        t = np.asarray(t).reshape(batch_shape + (3,))
    out[..., :3, :3] = R
    out[..., :3, 3] = t
    out[..., 3, 3] = 1.0
    return out


def rigid_transform_3d(A: torch.Tensor, B: torch.Tensor, weights: Optional[torch.Tensor] = None,
                       weight_threshold: float = 0) -> torch.Tensor:
    """utils.py:138-178: A, B [bs,K,3], weights [bs,K] -> [bs,4,4].  Like the reference this zeroes
    sub-threshold entries of a caller-supplied ``weights`` in place (utils.py:151)."""
    if weights is not None:
        weights[weights < weight_threshold] = 0
        weights = weights.contiguous().float()
    return ops.rigid_transform_3d_dense(A.contiguous().float(), B.contiguous().float(), weights, weight_threshold)


def transformation_error(pred_trans: torch.Tensor, gt_trans: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils.py:181-189: 4x4, 4x4 -> (RE degrees, TE) as 0-dim tensors."""
    re, te = ops.transformation_error_batched(pred_trans.reshape(1, 4, 4).contiguous().float(),
                                              gt_trans.reshape(1, 4, 4).contiguous().float())
    return re[0], te[0]


def register_from_prediction(src: torch.Tensor, src_pred: torch.Tensor, ref: torch.Tensor, s: float,
                             thresh: float) -> torch.Tensor:
    """models/pointnet.py:71-74: NN of src_pred/s in ref/s, Kabsch of (src[valid], ref[idx]) in the
    NORMALISED frame (no /s + c there).  Returns 4x4."""
    dev = src.device
    q = src_pred.reshape(-1, 3).contiguous().float()
    r = ref.reshape(-1, 3).contiguous().float()
    z, nq, nr = _i32([0], dev), _i32([q.shape[0]], dev), _i32([r.shape[0]], dev)
    idx, _, valid = ops.nn_search(q, r, z, nq, z, nr, torch.tensor([s], dtype=torch.float32, device=dev),
                                  q.shape[0], r.shape[0], thresh)
    T, _ = ops.kabsch_corr(src.reshape(-1, 3).contiguous().float(), r, z, nq, z, idx, valid,
                           torch.ones(1, device=dev), torch.zeros(1, 3, device=dev))
    return T[0]


def register_batch(batch: PackedBatch, src_pred: torch.Tensor, s: torch.Tensor, c: torch.Tensor,
                   dis_thresh: float, corr: str = "tgt"):
    """A7-A9 for every pair of a packed batch (evaluate_3d_match.py:94-101): thresholded 1-NN of
    src_pred in tgt, then Kabsch on the metric-frame correspondences.  s [B] fp32, c [B,3] fp32 (device).
    Returns (T [B,4,4], n_corr [B] int32, idx, dmin, valid) -- the last three indexed by packed source row."""
    B = batch.n_pairs
    tgt_xyz = batch.xyz[batch.rows_src:]
    tgt_row0 = batch.tgt_row0 - batch.rows_src
    idx, dmin, valid = ops.nn_search(src_pred, tgt_xyz, batch.src_row0, batch.src_len_dev, tgt_row0.contiguous(),
                                     batch.tgt_len_dev, s, max(batch.src_len), max(batch.tgt_len), dis_thresh)
    src_xyz = batch.xyz[: batch.rows_src]
    if corr == "tgt":
        T, n_corr = ops.kabsch_corr(src_xyz, tgt_xyz, batch.src_row0, batch.src_len_dev, tgt_row0.contiguous(),
                                    idx, valid, s, c)
    else:  # "src_pred": B = src_pred[valid] (evaluate_3d_match.py:99-101, 3DZeroMatch)
        T, n_corr = ops.kabsch_corr(src_xyz, src_pred, batch.src_row0, batch.src_len_dev, batch.src_row0, None,
                                    valid, s, c)
    return T, n_corr, idx, dmin, valid


def processbar(current, total):
    """Progress-bar string in the format of utils.py:17-23 (20 cells, "done / total")."""
    filled = min(20, int(20 * current / total))
    return "%s|   %d / %d" % ("\u2588" * filled + " " * (20 - filled), current, total)
