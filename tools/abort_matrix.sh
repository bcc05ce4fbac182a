#!/bin/bash
# GPU box: the C2 head under different prices of an insert for a streaming launch (cost model).
export TMPDIR=/tmp
for v in 120 300 600 1200; do
  echo "== GRP_T_ABORT_US=$v"
  GRP_T_ABORT_US=$v timeout 900 python3 bench.py --no-cpu-baseline --reads 1200000 --steps 4 --trace 2> /tmp/abort_$v.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['phases']['head']['seconds'], d['aux']['timed'])"
  grep "^reads" /tmp/abort_$v.err | python3 -c "
import sys,re
t=0
for l in sys.stdin:
    m=re.match(r'reads (\d+)\.\.(\d+): ([\d.]+) ms',l)
    if m and int(m.group(1))>=491520: t+=float(m.group(3))
print('ms behind read 491520:', round(t))"
done
