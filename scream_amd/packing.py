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
