#!/bin/bash
# Run on the GPU box (gpurun): PMC passes of the bench command, one counter group per run
# (never combined with sys/hip/hsa tracing).   tools/pmc_round.sh <tag>
# C2 geometry (G = 3e9: 65 GB of buckets, W = 60), 2 M reads: the insert-heavy head runs as
# batches (a few thousand launches), the rest as streaming windows; the summary keeps the
# k_query dispatches.
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
args="--no-cpu-baseline --no-pipeline-shaped --steps 4 --warmup 0 --reads 2000000"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/${tag}_pmcA -o a -- python3 bench.py $args > $out/${tag}_pmcA_bench.json 2> $out/${tag}_pmcA.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmcB -o b -- python3 bench.py $args > $out/${tag}_pmcB_bench.json 2> $out/${tag}_pmcB.err
# the write side (round 6: the fill is a read-modify-write; roofline_fill.traffic = read + write requests)
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $out/${tag}_pmcC -o c -- python3 bench.py $args > $out/${tag}_pmcC_bench.json 2> $out/${tag}_pmcC.err
A=$(find $out/${tag}_pmcA -name "*counter_collection.csv" | head -1)
B=$(find $out/${tag}_pmcB -name "*counter_collection.csv" | head -1)
Cc=$(find $out/${tag}_pmcC -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $out/${tag}_pmc_summary.json $out/${tag}_pmcA_bench.json $A $out/${tag}_pmcB_bench.json $B $out/${tag}_pmcC_bench.json $Cc
rm -rf $out/${tag}_pmcA $out/${tag}_pmcB $out/${tag}_pmcC
