#!/bin/bash
# round 5, GPU box: C4 geometry (h = 5, silver mode, read batches streamed through HBM) under the switches that change how a
# batch is confirmed: aux.counters must be identical
reads=${1:-300000}
sb=${2:-100000}
run() { env "$@" python3 bench.py --config C4 --reads $reads --stream-batch $sb --steps 4 --no-cpu-baseline --no-pipeline-shaped 2>/dev/null | grep '^{' | tail -1; }
run X=1 > /tmp/c1.json
run X=2 > /tmp/c2.json
run GRP_BATCH_OVERLAP=8 > /tmp/c3.json
run GRP_BATCH_VERIFY=off > /tmp/c4.json
run GRP_BATCH=off > /tmp/c5.json
python3 - <<'PY'
import json
r = [json.load(open("/tmp/c%d.json" % i)) for i in (1, 2, 3, 4, 5)]
for name, d in zip(("default #1        ", "default #2        ", "redo by the host  ", "second query      ", "no batches        "), r):
    print(name, round(d["value"]), "reads/s", d["aux"]["counters"], d["aux"]["timed"]["batches"], d["aux"]["batch_verify"]["flagged"])
same = all(d["aux"]["counters"] == r[0]["aux"]["counters"] for d in r)
print("IDENTICAL" if same else "DIFFERENT")
PY
