#!/bin/bash
# GPU box (round 3): batch / window thresholds with in-launch inserts (C2, first 2.5 M reads: head, transition, some steady state)
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
run() {
  name=$1; shift
  env "$@" timeout 600 python3 bench.py --reads 2500000 --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline-shaped > $out/r03_thr_$name.json 2> /dev/null
  python3 - $out/r03_thr_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["aux"]["timed"]
print(sys.argv[2], "wall %.3f s" % d["aux"]["wall_s"], {k: t[k] for k in ("windows", "reads_queried", "batches", "batches_undone", "stream_inserts")})
PY
}
run base A=1
run enter20 GRP_BATCH_ENTER=0.020 GRP_BATCH_LEAVE=0.012
run enter30 GRP_BATCH_ENTER=0.030 GRP_BATCH_LEAVE=0.018
run enter20_abort200 GRP_BATCH_ENTER=0.020 GRP_BATCH_LEAVE=0.012 GRP_T_ABORT_US=200
run enter30_abort150 GRP_BATCH_ENTER=0.030 GRP_BATCH_LEAVE=0.018 GRP_T_ABORT_US=150
run enter50_abort150 GRP_BATCH_ENTER=0.050 GRP_BATCH_LEAVE=0.030 GRP_T_ABORT_US=150
