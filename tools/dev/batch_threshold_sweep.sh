#!/bin/bash
# GPU box: C2 (3 M reads: head + the stretch where the insert rate falls through 1 %) against the
# insert rates at which the classifier enters / leaves the batches
for s in "GRP_BATCH_ENTER=0.012,GRP_BATCH_LEAVE=0.007" "GRP_BATCH_ENTER=0.006,GRP_BATCH_LEAVE=0.003" "GRP_BATCH_ENTER=0.02,GRP_BATCH_LEAVE=0.012" "GRP_BATCH_ENTER=0.004,GRP_BATCH_LEAVE=0.002"; do
  env $(echo $s | tr ',' ' ') python3 bench.py --no-cpu-baseline --reads 3000000 --steps 3 2>/dev/null > /tmp/bt.json
  python3 - "$s" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/bt.json") if l.startswith("{")][-1])
print(sys.argv[1].ljust(48), "reads/s", round(d["value"]), "batches", d["aux"]["timed"]["batches"], "batch_reads", d["aux"]["timed"]["batch_reads"])
PY
done
