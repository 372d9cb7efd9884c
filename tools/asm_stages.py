#!/usr/bin/env python3
"""Instruction mix between consecutive s_barrier instructions of one kernel (in program order): asm_stages.py file.s kernel-substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
segs = []; c = collections.Counter(); lab = 'entry'
for l in m.group(2).splitlines():
    t = l.strip()
    if not t or t.startswith(';'): continue
    if re.match(r'^\.LBB\d+_\d+:', t):
        c['labels'] += 1; lab = t.split(':')[0]; continue
    op = t.split()[0]
    if op == 's_barrier':
        segs.append((lab, c)); c = collections.Counter(); continue
    if op.startswith('v_mfma'): c['mfma'] += 1
    elif 'accvgpr' in op: c['acc_mov'] += 1
    elif op.startswith('v_pk_'): c['valu_pk'] += 1
    elif op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_log')): c['trans'] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op.startswith('ds_read') or op.startswith('ds_load'): c['ds_rd'] += 1
    elif op.startswith('ds_'): c['ds_wr'] += 1
    elif op.startswith('global_load_lds'): c['dma'] += 1
    elif op.startswith('global_load'): c['gld'] += 1
    elif op.startswith('global_store'): c['gst'] += 1
    elif op.startswith('scratch_'): c['scratch'] += 1
    elif op == 's_nop': c['nop'] += 1
segs.append((lab, c))
keys = ['mfma', 'valu', 'valu_pk', 'trans', 'acc_mov', 'ds_rd', 'ds_wr', 'dma', 'gld', 'gst', 'nop', 'scratch', 'labels']
print('%-4s %-10s ' % ('#', 'label') + ' '.join('%7s' % k for k in keys))
for i, (lab, c) in enumerate(segs):
    print('%-4d %-10s ' % (i, lab) + ' '.join('%7d' % c[k] for k in keys))
