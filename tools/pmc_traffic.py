#!/usr/bin/env python3
"""HBM traffic per kernel and per bench step from two rocprofv3 PMC passes (they cannot share a pass: FETCH_SIZE takes 3
of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/X/fetch -- python3 bench.py --steps 2 --warmup 1 --gen-procs 1 --no-cpu-baseline --lanes 1
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/X/write -- python3 bench.py ...same...
    python tools/pmc_traffic.py gpurun_out/X/fetch gpurun_out/X/write profiles/r02_traffic.json [--lanes 1]

Units and corrections as the guide prescribes: both counters are in KB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of wide (16 B per lane) coalesced reads -- every bulk read of these kernels, LDS-DMA included -- so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Steps are counted from the dispatches of pe_embed_ln_kernel
(one per lane and step).  bench.py reads the JSON (SCREAM_TRAFFIC_JSON or the newest profiles/r*_traffic.json) and puts
counter bytes beside the algorithmic bytes it computes itself."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter):
    agg = defaultdict(lambda: [0, 0.0])
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    assert files, "no *counter_collection.csv under %s" % d
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            e = agg[r["Kernel_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return agg


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    lanes = int(sys.argv[sys.argv.index("--lanes") + 1]) if "--lanes" in sys.argv else 1
    backend = sys.argv[sys.argv.index("--backend") + 1] if "--backend" in sys.argv else os.environ.get("SCREAM_GEMM", "h2")
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    steps_f = sum(v[0] for k, v in fe.items() if "pe_embed_ln_kernel" in k) / lanes
    steps_w = sum(v[0] for k, v in wr.items() if "pe_embed_ln_kernel" in k) / lanes
    assert steps_f > 0 and steps_w > 0, "pe_embed_ln_kernel not found in the passes"
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        nf, kb_f = fe.get(k, [0, 0.0])
        nw, kb_w = wr.get(k, [0, 0.0])
        fetch_b, write_b = kb_f * 1024.0 * 2.0, kb_w * 1024.0
        e = kernels.setdefault(short(k), {"dispatches_fetch_pass": 0, "dispatches_write_pass": 0, "fetch_bytes_per_step": 0.0,
                                           "write_bytes_per_step": 0.0})
        e["dispatches_fetch_pass"] += nf
        e["dispatches_write_pass"] += nw
        e["fetch_bytes_per_step"] += fetch_b / steps_f
        e["write_bytes_per_step"] += write_b / steps_w
    for e in kernels.values():
        e["fetch_bytes_per_launch"] = e["fetch_bytes_per_step"] * steps_f / max(e["dispatches_fetch_pass"], 1)
        e["write_bytes_per_launch"] = e["write_bytes_per_step"] * steps_w / max(e["dispatches_write_pass"], 1)
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py, summarised by tools/pmc_traffic.py",
           "corrections": "KB -> bytes; FETCH_SIZE x 2 (gfx950: 16 B/lane coalesced reads are reported at half); WRITE_SIZE exact",
           "lanes": lanes, "gemm_backend": backend, "steps_in_fetch_pass": steps_f, "steps_in_write_pass": steps_w,
           "fetch_bytes_per_step": sum(e["fetch_bytes_per_step"] for e in kernels.values()),
           "write_bytes_per_step": sum(e["write_bytes_per_step"] for e in kernels.values()),
           "kernels": {k: {kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in sorted(kernels.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_step"] + kv[1]["write_bytes_per_step"]))}}
    rec["hbm_bytes_per_step"] = rec["fetch_bytes_per_step"] + rec["write_bytes_per_step"]
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print("steps: %.1f / %.1f; HBM bytes per step: fetch %.3f GB + write %.3f GB" % (steps_f, steps_w, rec["fetch_bytes_per_step"] / 1e9, rec["write_bytes_per_step"] / 1e9))
    for k, v in list(rec["kernels"].items())[:8]:
        print("  %-28s fetch %8.1f MB/step  write %8.1f MB/step   (%.1f / %.1f MB per launch)" % (k[:28], v["fetch_bytes_per_step"] / 1e6, v["write_bytes_per_step"] / 1e6, v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
