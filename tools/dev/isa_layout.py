#!/usr/bin/env python3
"""developer: where the hot and the cold code of a kernel lie (instruction index of blocks with bucket loads / sleeps / host atomics)
   tools/dev/isa_layout.py <file.s> <mangled name prefix>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pre = sys.argv[2]
s = next(i for i, l in enumerate(lines) if l.startswith(pre) and not l.startswith('\t'))
e = next(i for i in range(s, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[s:e]
isins = lambda l: l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')
idx = 0
marks = []
for l in body:
    if isins(l):
        idx += 1
        t = l.strip()
        if 'global_load_dwordx4' in t: marks.append((idx, 'X4'))
        elif t.startswith('s_sleep'): marks.append((idx, 'sleep'))
        elif 'v_mad_u64_u32' in t or 'v_mul_hi_u32' in t: marks.append((idx, 'mul'))
        elif t.startswith('ds_add') or t.startswith('ds_cmpst') or t.startswith('ds_max'): marks.append((idx, 'ldsatomic'))
        elif t.startswith('s_barrier'): marks.append((idx, 'barrier'))
        elif 'global_atomic' in t: marks.append((idx, 'gatomic'))
print('total', idx)
# compress into ranges of 500 instructions
from collections import Counter
bins = {}
for i, k in marks:
    bins.setdefault(i // 500, Counter())[k] += 1
for b in sorted(bins):
    print('%6d..%6d' % (b * 500, b * 500 + 499), dict(bins[b]))
