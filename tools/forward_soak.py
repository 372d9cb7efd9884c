#!/usr/bin/env python3
"""Soak test of the whole forward under the lanes configuration: two sub-batches run A1-A10 concurrently on two
streams, over and over; every kernel is deterministic, so every repetition must reproduce the first one bit for bit.
usage: forward_soak.py [seconds] [x3|f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scream_amd import lanes
from scream_amd.geometry import register_batch
from scream_amd.model import PointTransformer
from scream_amd.packing import PackedBatch
from scream_amd.synthetic import make_state_dict
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
items = bench.make_items(list(range(12)), 6)
dev = torch.device("cuda:0")
net = PointTransformer(256, 6, 6); net.load_state_dict(make_state_dict(0, 256, 6, 6))
if len(sys.argv) > 2: net.gemm_backend = sys.argv[2]
net = net.to(dev).eval()
parts = []
for rg in lanes.split(len(items), 2):
    its = [items[i] for i in rg]
    b = PackedBatch.from_pairs([it[0].to(dev) for it in its], [it[1].to(dev) for it in its], [it[3].reshape(3).to(dev) for it in its])
    parts.append((b, torch.tensor([it[4] for it in its], device=dev), torch.stack([it[5] for it in its]).to(dev)))
def run(p):
    b, s, c = p
    pred = net.forward_packed(b)
    T, k, idx, dmin, valid = register_batch(b, pred, s, c, 0.1)
    return pred, T, idx
ref = lanes.run(dev, parts, run); torch.cuda.synchronize()
ref = [[t.clone() for t in r] for r in ref]
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    outs = lanes.run(dev, parts, run); torch.cuda.synchronize()
    for r, o in zip(ref, outs):
        for a, b in zip(r, o):
            if not torch.equal(a, b):
                print("MISMATCH at repetition", n, "max diff", (a.float() - b.float()).abs().max().item()); sys.exit(1)
    n += 1
print("forward soak ok: %d repetitions of 2 x %d pairs in %.0f s, bitwise stable (%s)" % (n, len(items) // 2, time.time() - t0, net.gemm_backend))
