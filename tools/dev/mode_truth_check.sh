#!/bin/bash
# GPU box: the first <reads> reads of a config (default C2: the real 61 Gbit filter, the whole
# insert-heavy head) through the product's default path (batches + streaming windows), twice,
# and through the classic windows only (GRP_BATCH=off): aux.counters and pop must be identical.
cfg=${1:-C2}
reads=${2:-700000}
run() { env "$@" python3 bench.py --config $cfg --reads $reads --steps 4 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1; }
run X=1 > /tmp/m1.json
run X=2 > /tmp/m2.json
run GRP_BATCH=off > /tmp/m3.json
python3 - <<'PY'
import json
r = [json.load(open("/tmp/m%d.json" % i)) for i in (1, 2, 3)]
for name, d in zip(("batches #1", "batches #2", "classic   "), r):
    print(name, round(d["value"]), "reads/s", d["aux"]["pop"], d["aux"]["counters"])
same = all((d["aux"]["pop"], d["aux"]["counters"]) == (r[0]["aux"]["pop"], r[0]["aux"]["counters"]) for d in r)
print("IDENTICAL" if same else "DIFFERENT")
PY
