#!/bin/bash
# GPU box: the insert-heavy head of C2 only (first <reads> reads), slice trace + per-kernel stats.
#   tools/head_profile.sh <tag> [reads]
tag=${1:-head}
reads=${2:-300000}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --reads $reads --steps 2 --warmup 0 --trace --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_head.json 2> $out/${tag}_head.err
grep "^reads" $out/${tag}_head.err | head -40
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 bench.py --reads $reads --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_head_rocprof.json 2> $out/${tag}_rocprof.err
find $out/${tag}_prof -name "*kernel_stats.csv" -exec cp {} $out/${tag}_head_kernel_stats.csv \;
rm -rf $out/${tag}_prof
cut -c1-150 $out/${tag}_head_kernel_stats.csv | head -25
