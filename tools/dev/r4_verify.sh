#!/bin/bash
# GPU box (round 4): grp_batch_verify on / off — the head of C2 (first <reads> reads) and the C4 geometry;
# the run's counters (hits, misses, IDs) must be identical in both modes
reads=${1:-300000}
c4reads=${2:-400000}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
show() {
  python3 - $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["aux"]["timed"]
ks = d["aux"].get("kernel_stats", {})
print(sys.argv[2], "batch_verify", d["aux"].get("batch_verify"))
print(sys.argv[2], "reads/s %.0f" % d["value"], "wall %.2f s" % d["aux"]["wall_s"], {k: t[k] for k in ("windows", "reads_queried", "inserts", "batches", "batches_undone", "batches_fused", "batch_reads") if k in t},
      "hits", d["aux"]["counters"]["hits"], "misses", d["aux"]["counters"]["misses"], "ids", d["aux"]["counters"]["ids_inserted"],
      {k: (v["launches"], round(v["ms"], 1)) for k, v in ks.items() if v.get("launches")})
PY
}
[ "$reads" = 0 ] || for m in on off; do
  GRP_BATCH_VERIFY=$m timeout 900 python3 bench.py --reads $reads --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/r04_head_verify_$m.json 2> $out/r04_head_verify_$m.err
  show $out/r04_head_verify_$m.json head_$m
done
for m in on off; do
  GRP_BATCH_VERIFY=$m timeout 900 python3 bench.py --config C4 --reads $c4reads --no-cpu-baseline > $out/r04_c4_verify_$m.json 2> $out/r04_c4_verify_$m.err
  show $out/r04_c4_verify_$m.json c4_$m
done
