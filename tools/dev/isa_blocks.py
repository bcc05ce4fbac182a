#!/usr/bin/env python3
"""developer: per basic block of a kernel in a --save-temps .s file: instruction mix (which blocks are the pass loop, what spills live there)
   tools/dev/isa_blocks.py <file.s> <mangled name prefix> [min x4 loads]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pre = sys.argv[2]
minx4 = int(sys.argv[3]) if len(sys.argv) > 3 else 8
s = next(i for i, l in enumerate(lines) if l.startswith(pre) and ':' in l[:len(pre) + 400] and not l.startswith('\t'))
e = next(i for i in range(s, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[s:e]
isins = lambda l: l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')
print(sys.argv[1], 'instructions', sum(1 for l in body if isins(l)), 'readlane', sum('v_readlane' in l for l in body), 'writelane', sum('v_writelane' in l for l in body))
labels = [i for i, l in enumerate(body) if re.match(r'\.LBB\d+_\d+:', l)]
labels.append(len(body))
for a, b in zip(labels, labels[1:]):
    blk = body[a:b]
    n4 = sum('global_load_dwordx4' in l for l in blk)
    if n4 >= minx4:
        ins = [l for l in blk if isins(l)]
        print(' block', body[a].split(':')[0], 'insts', len(ins), 'x4 loads', n4, 'readlane', sum('v_readlane' in l for l in ins), 'writelane', sum('v_writelane' in l for l in ins), 's_load', sum(l.strip().startswith('s_load') for l in ins),
              'scratch', sum('scratch_' in l for l in ins), 'ds', sum(l.strip().startswith('ds_') for l in ins), 's_waitcnt', sum('s_waitcnt' in l for l in ins), 's_nop', sum('s_nop' in l for l in ins))
