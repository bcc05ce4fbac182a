#!/bin/bash
# GPU box (round 3): the synchronous query forms of the batches with one frame per lane and the
# software-pipelined pass (GRP_SYNC_FR1: bit 0 plain query, bit 1 second query), C2 head and C4 geometry.
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for v in ${@:-0 3 1 2}; do
  GRP_SYNC_FR1=$v timeout 600 python3 bench.py --reads 400000 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/r03_fr1_c2_$v.json 2> $out/r03_fr1_c2_$v.err
  GRP_SYNC_FR1=$v timeout 600 python3 bench.py --config C4 --reads 400000 --steps 2 --warmup 0 --no-cpu-baseline > $out/r03_fr1_c4_$v.json 2> $out/r03_fr1_c4_$v.err
  for c in c2 c4; do
  python3 - $out/r03_fr1_${c}_$v.json $c $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = d["aux"]["timed"]
    print(sys.argv[2], "FR1=" + sys.argv[3], "reads/s %.0f" % d["value"], "wall %.2f s" % d["aux"]["wall_s"], "query_kernel_s %.2f" % d["aux"]["query_kernel_s"], {k: t[k] for k in ("reads_queried", "inserts", "batches", "batches_undone", "batches_fused")}, d["aux"]["counters"]["hits"], d["aux"]["counters"]["ids_inserted"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
