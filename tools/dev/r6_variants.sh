#!/bin/bash
# GPU box: side trees .ab_<name> (built beforehand) one after the other on the same box
#   tools/dev/r6_variants.sh <tag> "<name>:<env>;<name>:<env>..." [bench args]     (name "." = this tree)
tag=${1:-r06_var}
out=$PWD/gpurun_out
mkdir -p $out
IFS=';' read -ra sets <<< "$2"
i=0
for s in "${sets[@]}"; do
  i=$((i+1))
  name=${s%%:*}; envs=${s#*:}
  dir=.ab_$name; [ "$name" = "." ] && dir=.
  (cd $dir && env $envs timeout 300 python3 bench.py --no-cpu-baseline --no-pipeline-shaped "${@:3}" > $out/${tag}_$i.json 2> $out/${tag}_$i.err)
  python3 - "$out/${tag}_$i.json" "$s" <<'PY'
import json, sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    t=d["aux"]["timed"]
    print("[%s]" % sys.argv[2], "value %.0f" % d["value"], "steady %.0f" % (d["phases"]["steady"]["reads_per_s"] or 0), "head %.0f" % (d["phases"]["head"]["reads_per_s"] or 0), "frac %.4f" % d["roofline"]["frac"],
          "query ms %.0f" % d["aux"]["kernel_stats"]["query"]["ms"], "Gprobes/s %.2f" % d["aux"]["query_Gprobes_per_s"], "stream_inserts", t["stream_inserts"], "fallbacks", t["stream_insert_fallbacks"], "hits", d["aux"]["counters"]["hits"])
except Exception as e:
    print("[%s]" % sys.argv[2], "FAILED", e)
PY
done
