#!/bin/bash
# GPU box: the previous commit's tree (.ab_old, built beforehand) against this one on the same box
#   tools/dev/r6_old_new.sh <tag> [bench args]
tag=${1:-r06_on}
out=$PWD/gpurun_out
mkdir -p $out
summ() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    t=d["aux"]["timed"]
    print("[%s]" % sys.argv[2], "value %.0f" % d["value"], "steady %.0f" % (d["phases"]["steady"]["reads_per_s"] or 0), "head %.0f" % (d["phases"]["head"]["reads_per_s"] or 0), "frac %.4f" % d["roofline"]["frac"],
          "query ms %.0f" % d["aux"]["kernel_stats"]["query"]["ms"], "Gprobes/s %.2f" % d["aux"]["query_Gprobes_per_s"], "stream_inserts", t["stream_inserts"], "fallbacks", t["stream_insert_fallbacks"], "hits", d["aux"]["counters"]["hits"])
except Exception as e:
    print("[%s]" % sys.argv[2], "FAILED", e)
PY
}
(cd .ab_old && timeout 300 python3 bench.py --no-cpu-baseline --no-pipeline-shaped "${@:2}" > $out/${tag}_old1.json 2> $out/${tag}_old1.err); summ $out/${tag}_old1.json old
i=0
for s in "GRP_STREAM_KEEP=0" "GRP_STREAM_KEEP=0 GRP_STREAM_PASS_LOOK=off" "GRP_STREAM_KEEP=2" "GRP_STREAM_KEEP=2 GRP_STREAM_PASS_LOOK=off"; do
  i=$((i+1))
  env $s timeout 300 python3 bench.py --no-cpu-baseline --no-pipeline-shaped "${@:2}" > $out/${tag}_new$i.json 2> $out/${tag}_new$i.err; summ $out/${tag}_new$i.json "$s"
done
(cd .ab_old && timeout 300 python3 bench.py --no-cpu-baseline --no-pipeline-shaped "${@:2}" > $out/${tag}_old2.json 2> $out/${tag}_old2.err); summ $out/${tag}_old2.json old
