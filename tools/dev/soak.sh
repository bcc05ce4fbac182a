#!/bin/bash
# GPU box: the default workload <n> times in a row — do the runs end with the same counters (a race in the
# in-launch insert protocol or the batches would show as a difference or a hang)?   tools/dev/soak.sh [n] [config]
n=${1:-5}
cfg=${2:-C2}
export TMPDIR=/tmp
for i in $(seq 1 $n); do
  timeout 900 python3 bench.py --config $cfg --no-cpu-baseline --no-pipeline-shaped 2>/dev/null | grep '^{' | tail -1 > /tmp/soak_$i.json
done
python3 - $n <<'PY'
import json, sys
n = int(sys.argv[1])
r = [json.load(open("/tmp/soak_%d.json" % i)) for i in range(1, n + 1)]
for i, d in enumerate(r):
    t = d["aux"]["timed"]
    print("run %d  %.0f reads/s  head %.2f s  steady %.0f  stream_inserts %d  batches %d (%d undone)  pop %d  hits %d  misses %d  ids %d" % (
        i + 1, d["value"], d["phases"]["head"]["seconds"], d["phases"]["steady"]["reads_per_s"], t["stream_inserts"], t["batches"], t["batches_undone"], d["aux"]["pop"],
        d["aux"]["counters"]["hits"], d["aux"]["counters"]["misses"], d["aux"]["counters"]["ids_inserted"]))
same = all((d["aux"]["pop"], d["aux"]["counters"]) == (r[0]["aux"]["pop"], r[0]["aux"]["counters"]) for d in r)
print("IDENTICAL" if same else "DIFFERENT")
PY
