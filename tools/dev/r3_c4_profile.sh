#!/bin/bash
# GPU box: per-kernel time of the C4 geometry (h = 5, silver mode: the stream is head after head)
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_c4prof -o c4 -- python3 bench.py --config C4 --reads 600000 --steps 2 --warmup 0 --no-cpu-baseline > $out/r03_c4prof.json 2> $out/r03_c4prof.err
find $out/r03_c4prof -name "*kernel_stats.csv" -exec cp {} $out/r03_c4_head_kernel_stats.csv \;
rm -rf $out/r03_c4prof
cut -c1-110 $out/r03_c4_head_kernel_stats.csv | head -16
python3 -c "
import json
d=json.loads(open('$out/r03_c4prof.json').read().strip().splitlines()[-1]); print(d['value'], d['aux']['wall_s'], d['aux']['timed'])"
