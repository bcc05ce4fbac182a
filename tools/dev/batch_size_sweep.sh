#!/bin/bash
# GPU box: the first 300 k reads of C2 (the insert-heavy head) with fixed batch sizes against the
# adaptive choice (batch_feedback)
for s in "X=adaptive" "GRP_BATCH_READS=96" "GRP_BATCH_READS=160" "GRP_BATCH_READS=256" "GRP_BATCH_READS=400"; do
  env $s python3 bench.py --no-cpu-baseline --reads 300000 --steps 2 --warmup 0 2>/dev/null > /tmp/bs.json
  python3 - "$s" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/bs.json") if l.startswith("{")][-1])
t = d["aux"]["timed"]
print(sys.argv[1].ljust(22), "reads/s", round(d["value"]), "batches", t["batches"], "ended early", t["batches_undone"])
PY
done
