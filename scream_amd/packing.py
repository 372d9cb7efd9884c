"""Var-len packing of B registration pairs into one row stream.

The reference runs one pair per forward (models/pointnet.py:39-40 asserts B == 1) because N and M
differ per pair.  Here B pairs share every kernel launch: rows are laid out
[src cloud 0 | ... | src cloud B-1 | tgt cloud 0 | ... | tgt cloud B-1], each cloud starting on a
128-row boundary and zero padded to one, so a 128-row kernel tile never spans two clouds and the
per-cloud reductions (K^T V, Ksum, v / S) stay per pair.  Cloud ids: source of pair p = p, target = B + p.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from .ops import KV_CHUNK, ROW_TILE


class StagingBuffers:
    """Two alternating pairs of pinned host buffers (float32, int32): the asynchronous upload of batch i may still be
    reading one pair while batch i+1 is packed into the other; a pair is reused only after its upload event."""

    def __init__(self):
        self._slots = [[None, None, None], [None, None, None]]
        self._next = 0

    def take(self, n_float: int, n_int: int):
        slot = self._slots[self._next]
        self._next ^= 1
        if slot[2] is not None:
            slot[2].synchronize()
        if slot[0] is None or slot[0].numel() < n_float:
            slot[0] = torch.empty(max(n_float, 1) * 5 // 4, dtype=torch.float32, pin_memory=torch.cuda.is_available())
        if slot[1] is None or slot[1].numel() < n_int:
            slot[1] = torch.empty(max(n_int, 1) * 5 // 4, dtype=torch.int32, pin_memory=torch.cuda.is_available())
        if torch.cuda.is_available():
            slot[2] = torch.cuda.Event()
        self._last = slot
        return slot[0], slot[1]

    def uploaded(self):
        """Call after the copies that read the last taken buffers have been enqueued (on the current stream)."""
        if torch.cuda.is_available() and self._last[2] is not None:
            self._last[2].record()


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


@dataclass
class PackedBatch:
    n_pairs: int
    src_len: List[int]
    tgt_len: List[int]
    cloud_row0_host: np.ndarray  # int32 [2B]
    cloud_len_host: np.ndarray   # int32 [2B]
    rows_src: int
    rows_total: int
    max_chunks: int
    xyz: torch.Tensor          # [rows_total, 3]
    center: torch.Tensor       # [2B, 3]
    tile_cloud: torch.Tensor   # int32 [rows_total / 128]
    cloud_row0: torch.Tensor   # int32 [2B]
    cloud_len: torch.Tensor    # int32 [2B]

    @staticmethod
    def layout(src_len: Sequence[int], tgt_len: Sequence[int]):
        B = len(src_len)
        lens = np.array(list(src_len) + list(tgt_len), dtype=np.int32)
        if (lens <= 0).any():
            raise ValueError("every cloud needs at least one point")
        padded = (lens.astype(np.int64) + ROW_TILE - 1) // ROW_TILE * ROW_TILE
        row0 = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
        rows_src = int(padded[:B].sum())
        rows_total = int(padded.sum())
        if rows_total >= 2 ** 31:
            raise ValueError("batch too large for int32 row indices")
        tile_cloud = np.repeat(np.arange(2 * B, dtype=np.int32), (padded // ROW_TILE).astype(np.int64))
        max_chunks = int(((lens.max() + KV_CHUNK - 1) // KV_CHUNK))
        return lens, row0.astype(np.int32), rows_src, rows_total, tile_cloud, max_chunks

    @classmethod
    def from_pairs(cls, srcs: Sequence[torch.Tensor], tgts: Sequence[torch.Tensor],
                   centers: Optional[Sequence[Optional[torch.Tensor]]] = None) -> "PackedBatch":
        """srcs[i] [N_i,3], tgts[i] [M_i,3] fp32 device tensors; centers[i] [3] or None (= mean of src,
        models/pointnet.py:43-44)."""
        B = len(srcs)
        assert B == len(tgts) and B > 0
        dev = srcs[0].device
        src_len = [int(t.shape[0]) for t in srcs]
        tgt_len = [int(t.shape[0]) for t in tgts]
        lens, row0, rows_src, rows_total, tile_cloud, max_chunks = cls.layout(src_len, tgt_len)
        xyz = torch.zeros(rows_total, 3, device=dev, dtype=torch.float32)
        center = torch.zeros(2 * B, 3, device=dev, dtype=torch.float32)
        for i, t in enumerate(list(srcs) + list(tgts)):
            xyz[int(row0[i]):int(row0[i]) + int(lens[i])] = t.reshape(-1, 3).to(torch.float32)
        for i in range(B):
            c = None if centers is None else centers[i]
            center[i] = srcs[i].reshape(-1, 3).mean(dim=0) if c is None else c.reshape(3).to(torch.float32)
        return cls(B, src_len, tgt_len, row0, lens, rows_src, rows_total, max_chunks, xyz, center,
                   torch.from_numpy(tile_cloud).to(dev), torch.from_numpy(row0).to(dev),
                   torch.from_numpy(lens).to(dev))

    @classmethod
    def from_host(cls, srcs: Sequence[torch.Tensor], tgts: Sequence[torch.Tensor], centers: Sequence[torch.Tensor],
                  device: torch.device, extra: Optional[np.ndarray] = None, staging=None):
        """Pack CPU clouds on the host and upload the batch with TWO asynchronous copies from pinned memory (floats:
        packed xyz | centres | `extra`; int32: tile_cloud | cloud_row0 | cloud_len) on the current stream, instead of
        one blocking copy per cloud.  Returns (batch, extra_on_device).  `staging` is an optional StagingBuffers that
        recycles the pinned memory."""
        B = len(srcs)
        assert B == len(tgts) == len(centers) and B > 0
        src_len = [int(t.shape[0]) for t in srcs]
        tgt_len = [int(t.shape[0]) for t in tgts]
        lens, row0, rows_src, rows_total, tile_cloud, max_chunks = cls.layout(src_len, tgt_len)
        n_extra = 0 if extra is None else int(extra.size)
        nf = rows_total * 3 + 2 * B * 3 + n_extra
        ni = tile_cloud.size + 4 * B
        fbuf, ibuf = (staging or StagingBuffers()).take(nf, ni)
        f = fbuf.numpy()
        f[: rows_total * 3] = 0.0
        xyz_h = f[: rows_total * 3].reshape(rows_total, 3)
        for i, t in enumerate(list(srcs) + list(tgts)):
            xyz_h[int(row0[i]):int(row0[i]) + int(lens[i])] = t.detach().reshape(-1, 3).float().cpu().numpy()
        cen = f[rows_total * 3: rows_total * 3 + 6 * B].reshape(2 * B, 3)
        cen[:] = 0.0
        for i in range(B):
            cen[i] = centers[i].detach().reshape(3).float().cpu().numpy()
        if n_extra:
            f[rows_total * 3 + 6 * B: nf] = np.asarray(extra, dtype=np.float32).reshape(-1)
        iv = ibuf.numpy()
        iv[: tile_cloud.size] = tile_cloud
        iv[tile_cloud.size: tile_cloud.size + 2 * B] = row0
        iv[tile_cloud.size + 2 * B: ni] = lens
        fd = fbuf[:nf].to(device, non_blocking=True)
        idv = ibuf[:ni].to(device, non_blocking=True)
        batch = cls(B, src_len, tgt_len, row0, lens, rows_src, rows_total, max_chunks, fd[: rows_total * 3].view(rows_total, 3),
                    fd[rows_total * 3: rows_total * 3 + 6 * B].view(2 * B, 3), idv[: tile_cloud.size],
                    idv[tile_cloud.size: tile_cloud.size + 2 * B], idv[tile_cloud.size + 2 * B: ni])
        return batch, fd[rows_total * 3 + 6 * B: nf]

    # ---- views used by the search / solve stage ------------------------------------------------
    @property
    def src_row0(self) -> torch.Tensor:
        return self.cloud_row0[: self.n_pairs]

    @property
    def src_len_dev(self) -> torch.Tensor:
        return self.cloud_len[: self.n_pairs]

    @property
    def tgt_row0(self) -> torch.Tensor:
        return self.cloud_row0[self.n_pairs:]

    @property
    def tgt_len_dev(self) -> torch.Tensor:
        return self.cloud_len[self.n_pairs:]

    def unpack_src(self, packed: torch.Tensor) -> List[torch.Tensor]:
        """Split a [rows_src, ...] tensor into per-pair [N_i, ...] views."""
        return [packed[int(self.cloud_row0_host[i]):int(self.cloud_row0_host[i]) + self.src_len[i]]
                for i in range(self.n_pairs)]
