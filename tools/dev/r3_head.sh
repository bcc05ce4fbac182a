#!/bin/bash
# GPU box (round 3): the insert-heavy head of C2 (first <reads> reads), fused batch queries on / off
reads=${1:-400000}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for m in on off; do
  GRP_BATCH_FUSE=$m timeout 600 python3 bench.py --reads $reads --steps 2 --warmup 0 --trace --no-cpu-baseline --no-pipeline-shaped > $out/r03_head_$m.json 2> $out/r03_head_$m.err
  python3 - $out/r03_head_$m.json $m <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["aux"]["timed"]
print(sys.argv[2], "reads/s %.0f" % d["value"], "wall %.2f s" % d["aux"]["wall_s"], {k: t[k] for k in ("windows", "reads_queried", "inserts", "batches", "batches_undone", "batches_fused", "batch_reads")}, d["aux"]["counters"]["hits"], d["aux"]["counters"]["ids_inserted"])
PY
done
