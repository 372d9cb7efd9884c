#!/usr/bin/env python3
"""How well the vector instructions of a kernel are interleaved with its matrix instructions: histogram of the lengths of the runs of
VALU-class instructions between two consecutive MFMAs (and of the MFMA runs with nothing between them): asm_runs.py file.s kernel-substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
vr = collections.Counter(); mr = collections.Counter(); v = 0; mm = 0; seen = False; cyc = 0
for l in m.group(2).splitlines():
    t = l.strip()
    if not t or t.startswith(';') or t.endswith(':'): continue
    op = t.split()[0]
    if op.startswith('v_mfma'):
        if seen: vr[min(v, 40)] += 1
        if v == 0: mm += 1
        else:
            if mm: mr[mm] += 1
            mm = 1
        v = 0; seen = True
    elif op.startswith('v_'):
        v += 1
print("VALU run length between consecutive MFMAs -> count:", sorted(vr.items()))
print("back-to-back MFMA run length -> count:", sorted(mr.items()))
