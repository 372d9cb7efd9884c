"""Concurrent lanes: a batch of pairs split into L independent sub-batches, each on its own HIP stream.

Pairs never interact, so nothing orders lane 0's kernels against lane 1's.  The GEMM kernels
are persistent grids of one block per CU; with 650 or 1302 output tiles a launch ends in a partial round during
which most CUs idle, and every launch ends with a store-bound epilogue.  A second lane's kernels fill exactly those
holes (measured: 37.7 -> 35.7 ms per 32-pair forward with two lanes, 35.4 with four).  Same kernels, same inputs per
pair, same results bit for bit -- only the interleaving on the chip changes.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, TypeVar

import torch

T = TypeVar("T")
DEFAULT_LANES = 2
_streams = {}


def lane_streams(device: torch.device, n: int) -> List[torch.cuda.Stream]:
    pool = _streams.setdefault((device.type, device.index), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def split(n_items: int, lanes: int) -> List[range]:
    """Contiguous, near-equal index ranges (never empty): ceil-sized lanes first."""
    lanes = max(1, min(lanes, n_items))
    q, r = divmod(n_items, lanes)
    out, lo = [], 0
    for i in range(lanes):
        hi = lo + q + (1 if i < r else 0)
        out.append(range(lo, hi))
        lo = hi
    return out


def split_weighted(weights: Sequence[float], lanes: int) -> List[range]:
    """Contiguous index ranges (never empty) with near-equal total weight: lane l ends where the running sum is closest
    to (l + 1) / lanes of the total.  A lane's time follows its point count, so real batches (3DMatch clouds range
    from 2 k to 10 k points) are split by weight = N_i + M_i rather than by pair count."""
    n = len(weights)
    lanes = max(1, min(lanes, n))
    total = float(sum(weights))
    if lanes == 1 or total <= 0:
        return split(n, lanes)
    prefix, acc = [], 0.0
    for w in weights:
        acc += float(w)
        prefix.append(acc)
    cuts, lo = [], 0
    for l in range(1, lanes):
        target = total * l / lanes
        # candidates lo+1 .. n-(lanes-l): keep at least one item for this lane and for every lane after it
        best = min(range(lo + 1, n - (lanes - l) + 1), key=lambda j: abs(prefix[j - 1] - target))
        cuts.append(best)
        lo = best
    bounds = [0] + cuts + [n]
    return [range(a, b) for a, b in zip(bounds, bounds[1:])]


def run(device: torch.device, parts: Sequence, fn: Callable[[object], T]) -> List[T]:
    """fn(part) for every part, part i on lane stream i, forked from and joined to the current stream.  Tensors that
    fn allocates belong to the lane stream's pool; the join makes them safe to read on the current stream, and callers
    consume them (or copy them to the host) before the next call reuses the lanes."""
    if len(parts) == 1:
        return [fn(parts[0])]
    cur = torch.cuda.current_stream(device)
    outs = []
    streams = lane_streams(device, len(parts))
    for st, part in zip(streams, parts):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(fn(part))
    for st in streams:
        cur.wait_stream(st)
    return outs
