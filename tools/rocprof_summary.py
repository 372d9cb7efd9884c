#!/usr/bin/env python3
"""Condense a rocprofv3 output directory into the small text summaries committed under profiles/.

    python tools/rocprof_summary.py <rocprof_dir> <out_prefix>

Writes <out_prefix>_kernel_stats.txt from *kernel_stats.csv (or aggregates *kernel_trace.csv itself), and
<out_prefix>_pmc.txt with per-kernel means of every counter found in *counter_collection.csv."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name, n=90):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    d, out = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    traces = find(d, "*kernel_trace.csv")
    if traces:
        agg = defaultdict(list)
        for f in traces:
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        total = sum(sum(v) for v in agg.values())
        with open(out + "_kernel_stats.txt", "w") as fo:
            fo.write("# from %s (rocprofv3 --kernel-trace); durations in microseconds\n" % ", ".join(os.path.basename(t) for t in traces))
            fo.write("%-92s %8s %12s %10s %10s %10s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                fo.write("%-92s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%\n" % (short(k), len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / total))
        print("wrote", out + "_kernel_stats.txt")
    pmcs = find(d, "*counter_collection.csv")
    if pmcs:
        agg = defaultdict(lambda: defaultdict(list))
        for f in pmcs:
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(out + "_pmc.txt", "w") as fo:
            fo.write("# from %s (rocprofv3 --pmc); mean per dispatch\n" % ", ".join(os.path.basename(t) for t in pmcs))
            for k, cs in sorted(agg.items()):
                fo.write("%s\n" % short(k, 120))
                for c, v in sorted(cs.items()):
                    fo.write("    %-28s dispatches %6d  mean %.6g\n" % (c, len(v), sum(v) / len(v)))
        print("wrote", out + "_pmc.txt")


if __name__ == "__main__":
    main()
