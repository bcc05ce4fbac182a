#!/bin/bash
# GPU box: the first <reads> reads of a config (default C2: the real 61 Gbit filter, the whole
# insert-heavy head) through the product's default path (batches + streaming windows), twice,
# through the classic windows only (GRP_BATCH=off), through round 2's forms (no in-launch inserts, no
# fused batch queries), with every seed hashed on its own and through round 3's batches (second query instead of the
# patch from the records): aux.counters and pop must be identical.
cfg=${1:-C2}
reads=${2:-700000}
run() { env "$@" python3 bench.py --config $cfg --reads $reads --steps 4 --no-cpu-baseline --no-pipeline-shaped 2>/dev/null | grep '^{' | tail -1; }
run X=1 > /tmp/m1.json
run X=2 > /tmp/m2.json
run GRP_BATCH=off > /tmp/m3.json
run GRP_STREAM_RESUME=off GRP_BATCH_FUSE=off > /tmp/m4.json   # round 2's forms: windows end at inserts, batches query twice
run GRP_SHARED_HALVES=off > /tmp/m5.json                      # every seed hashed on its own
run GRP_BATCH_VERIFY=off > /tmp/m6.json   # round 3's forms: the batch's second decisions by a second query of every read
run GRP_STREAM_KEEP=0 GRP_BATCH_OVERLAP=8 > /tmp/m7.json   # streaming windows that keep nothing across an insert (rounds 3 - 5), round 4's fixed overlap threshold
python3 - <<'PY'
import json
r = [json.load(open("/tmp/m%d.json" % i)) for i in (1, 2, 3, 4, 5, 6, 7)]
for name, d in zip(("default #1      ", "default #2      ", "no batches      ", "round-2 forms   ", "no shared halves", "round-3 forms   ", "keep nothing    "), r):
    print(name, round(d["value"]), "reads/s", d["aux"]["pop"], d["aux"]["counters"])
same = all((d["aux"]["pop"], d["aux"]["counters"]) == (r[0]["aux"]["pop"], r[0]["aux"]["counters"]) for d in r)
print("IDENTICAL" if same else "DIFFERENT")
PY
