#!/usr/bin/env python3
"""Experiment: the forward of a 32-pair step as L independent lanes (pairs are independent) on L HIP streams, so
that one lane's partial last rounds / epilogues are filled by the other lanes' kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scream_amd.model import PointTransformer
from scream_amd.packing import PackedBatch
from scream_amd.synthetic import make_state_dict
items = bench.make_items(list(range(32)), 8)
dev = torch.device("cuda:0")
sd = make_state_dict(0, 256, 6, 6)
def mk():
    n = PointTransformer(256, 6, 6); n.load_state_dict(sd); return n.to(dev).eval()
for L in (1, 2, 4, 1, 2, 4):
    nets = [mk() for _ in range(L)]
    streams = [torch.cuda.Stream() for _ in range(L)]
    per = 32 // L
    batches = [PackedBatch.from_pairs([it[0].to(dev) for it in items[l * per:(l + 1) * per]], [it[1].to(dev) for it in items[l * per:(l + 1) * per]],
                                      [it[3].reshape(3).to(dev) for it in items[l * per:(l + 1) * per]]) for l in range(L)]
    def step():
        outs = []
        for l in range(L):
            with torch.cuda.stream(streams[l]):
                outs.append(nets[l].forward_packed(batches[l]))
        return outs
    torch.cuda.synchronize()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("lanes %d: forward %.2f ms per 32 pairs -> %.0f pairs/s (forward only)" % (L, dt * 1e3, 32 / dt), flush=True)
