#!/usr/bin/env python3
"""Time of the batched GPU ICP (scream_icp_p2p) with the target grid of csrc/icp_grid.hip against the brute-force search
(SCREAM_ICP_BRUTE=1): 32 3DMatch-like pairs (radius 0.1 m, 30 iterations) and 8 KITTI-like pairs (radius 0.6 m, 200 iterations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import scream_ref as O  # ground-truth poses of the synthetic pairs only (a tool, not the product path)
from scream_amd import ops
from scream_amd.data import SyntheticPairs, normalize_pair
from scream_amd.packing import PackedBatch
from scream_amd.synthetic import make_kitti_pair
DEV = "cuda:0"


def case(kind):
    if kind == "3dmatch":
        items = [SyntheticPairs("3dmatch", 32, seed0=300)[i] for i in range(32)]
        radius, iters = 0.1, 30
    else:
        items = []
        for j in range(8):
            src, tgt, rot, trans, s_, c_ = normalize_pair(*make_kitti_pair(40 + j), "bbox")
            items.append((src, tgt, rot, trans, s_, None, None, c_))
        radius, iters = 0.6, 200
    batch = PackedBatch.from_pairs([it[0].to(DEV) for it in items], [it[1].to(DEV) for it in items], None)
    s = torch.tensor([it[4] for it in items], dtype=torch.float32, device=DEV)
    c = torch.stack([it[7] for it in items]).to(DEV)
    rng = np.random.default_rng(1)
    T0 = []
    for it in items:
        Tgt = O.gt_pose_metric(it[2], it[3], it[4], it[7]).double().numpy()
        ang = np.radians(1.5)
        P = np.eye(4)
        P[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        P[:3, 3] = rng.normal(scale=0.02, size=3)
        T0.append(P @ Tgt)
    T0 = torch.from_numpy(np.stack(T0)).float().to(DEV)
    tgt_row0 = (batch.tgt_row0 - batch.rows_src).contiguous()
    args = (batch.xyz[: batch.rows_src], batch.xyz[batch.rows_src:], batch.src_row0, batch.src_len_dev, tgt_row0,
            batch.tgt_len_dev, s, c, T0, max(batch.src_len), max(batch.tgt_len), radius, iters)
    out = {}
    for mode in ("grid", "brute"):
        if mode == "brute": os.environ["SCREAM_ICP_BRUTE"] = "1"
        else: os.environ.pop("SCREAM_ICP_BRUTE", None)
        res = ops.icp_p2p(*args); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.0:
            res = ops.icp_p2p(*args); torch.cuda.synchronize(); n += 1
        out[mode] = ((time.perf_counter() - t0) / n * 1e3, res)
    os.environ.pop("SCREAM_ICP_BRUTE", None)
    same = all(torch.equal(a, b) for a, b in zip(out["grid"][1], out["brute"][1]))
    it_ = out["grid"][1][2].float().mean().item()
    print("%-8s %2d pairs, mean %5.0f / %5.0f points, radius %.1f m, %.1f iterations on average: grid %.2f ms, brute force %.2f ms "
          "(%.1fx), identical results: %s" % (kind, len(items), np.mean(batch.src_len), np.mean(batch.tgt_len), radius, it_,
                                             out["grid"][0], out["brute"][0], out["brute"][0] / out["grid"][0], same))


if __name__ == "__main__":
    case("3dmatch")
    case("kitti")
