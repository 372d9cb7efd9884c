#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in an assembly file: asm_mix.py file.s kernel-substring [min_mfma]"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
body = m.group(2).splitlines()
blocks = []; cur = ['entry', collections.Counter()]
for l in body:
    t = l.strip()
    if not t or t.startswith(';'): continue
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append(cur); cur = [t.split(':')[0], collections.Counter()]; continue
    op = t.split()[0]; c = cur[1]
    if op.startswith('v_mfma'): c['mfma'] += 1
    elif 'accvgpr' in op: c['acc_mov'] += 1
    elif op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): c['lane'] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op.startswith('ds_'): c['ds'] += 1
    elif op.startswith('global_'): c['vmem'] += 1
    elif op.startswith('scratch_'): c['scratch'] += 1
    elif op == 's_barrier': c['barrier'] += 1
    elif op == 's_waitcnt': c['wait'] += 1
    elif op == 's_nop': c['nop'] += 1
    elif op.startswith('s_'): c['salu'] += 1
blocks.append(cur)
tot = collections.Counter()
for n, c in blocks:
    tot.update(c)
    if c['mfma'] >= lim or c['valu'] > 150: print(n, dict(c))
print('total', dict(tot))
