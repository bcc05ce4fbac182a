#!/bin/bash
# round 6, GPU box: is the far table what holds C1's silver pass back?  The same reads through
# silver mode with the miBF at 1 / 2 / 4 / 8 x the size -x gives (occupancy 0.89 -> ~0.1): the share of the count words
# that live in the far table, the collect pass's record slots per second, reads per second; and the pass's kernel stats.
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
tag=${1:-r06_c1_far}
for sc in 1 2 4 8; do
  timeout 600 python3 bench.py --config C1 --silver 5 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped --filter-scale $sc > $out/${tag}_x$sc.json 2> $out/${tag}_x$sc.err
  python3 - $out/${tag}_x$sc.json $sc <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ks = d["aux"]["kernel_stats"]; bv = d["aux"]["batch_verify"]; ri = d.get("roofline_insert") or {}
    print("x%s" % sys.argv[2], "occupancy %.3f" % d["config"]["occupancy"], "reads/s %.0f" % d["value"], "far count words", bv["far_count_words"], "share of pop %.3f" % (bv["far_count_words"] / d["config"]["pop"]),
          "collect+apply G slots/s %.2f" % (ri.get("G_record_slots_per_s") or 0), "kernel ms: query %.0f verify %.0f batch_insert %.0f decide %.0f" % (ks["query"]["ms"], ks["verify"]["ms"], ks["batch_insert"]["ms"], ks["decide"]["ms"]),
          "wall %.2f s" % d["aux"]["wall_s"], "batches", d["aux"]["timed"]["batches"], "undone", d["aux"]["timed"].get("batches_undone"))
except Exception as e:
    print("x%s" % sys.argv[2], "FAILED", e)
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o sp -- python3 bench.py --config C1 --silver 5 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_under_rocprof.json 2> /dev/null
f=$(find $out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp $f $out/${tag}_silver_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-64s %7s calls %9.1f ms  avg %8.3f ms  %4.1f %%" % (r["Name"].split("(")[0][-64:], r["Calls"], int(r["TotalDurationNs"]) / 1e6, int(r["TotalDurationNs"]) / 1e6 / int(r["Calls"]), 100.0 * int(r["TotalDurationNs"]) / tot))
PY
rm -rf $out/${tag}_prof
