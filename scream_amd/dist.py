"""Data-parallel sharding of registration pairs.

Pairs are independent units: rank r of W takes pair indices r, r+W, ...; every rank holds the full
57 MB weight replica.  The ONLY exchange is one all-gather of a fixed-width per-pair metric row at the
end of an evaluation (RCCL over xGMI through torch.distributed's "nccl" backend on the GPUs, "gloo" in
the CPU tests).  Rows, not sums, are gathered because the per-scene median needs the full lists
(evaluate_3d_match.py:152-160).  The payload is ~32 B per pair, i.e. latency-bound.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

ROW_WIDTH = 8  # pair_id, scene_idx, counted, success, re, te, rmse, point_loss
COL_PAIR, COL_SCENE, COL_COUNTED, COL_SUCCESS, COL_RE, COL_TE, COL_RMSE, COL_LOSS = range(ROW_WIDTH)


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun's contract).
    Returns (rank, world, local_rank).  A single-process run needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Round-robin: balances the scene mix and (N + M) across ranks without any metadata exchange."""
    return list(range(rank, n, world))


def all_gather_rows(rows: np.ndarray, device: Optional[torch.device] = None) -> np.ndarray:
    """rows [k, ROW_WIDTH] float64 on this rank -> [sum k, ROW_WIDTH] on every rank, ordered by pair id.
    Ranks may hold different k: rows are padded to the max count with pair_id = -1 and dropped afterwards."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, ROW_WIDTH)
    rank, world = rank_world()
    if world == 1:
        out = rows
    else:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        kmax = int(cnt.item())
        padded = np.full((kmax, ROW_WIDTH), -1.0, dtype=np.float64)
        padded[: rows.shape[0]] = rows
        mine = torch.from_numpy(padded).to(device)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        out = torch.cat(gathered, dim=0).cpu().numpy()
        out = out[out[:, COL_PAIR] >= 0]
    return out[np.argsort(out[:, COL_PAIR], kind="stable")]
