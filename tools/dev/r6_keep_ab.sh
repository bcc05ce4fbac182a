#!/bin/bash
# GPU box: the in-launch insert with and without the kept tiles (round 6, GRP_STREAM_KEEP = fingerprint buffers per workgroup)
#   tools/dev/r6_keep_ab.sh <tag> "<env settings>;<env settings>;..." [bench args]
tag=${1:-r06}
out=gpurun_out
mkdir -p $out
IFS=';' read -ra sets <<< "$2"
i=0
for s in "${sets[@]}"; do
  i=$((i+1))
  env GRP_TRACE_ABORT=1 $s timeout 300 python3 bench.py --no-cpu-baseline --no-pipeline-shaped "${@:3}" > $out/${tag}_$i.json 2> $out/${tag}_$i.err
  python3 - "$out/${tag}_$i.json" "$s" <<'PY'
import json, sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    t=d["aux"]["timed"]
    print("[%s]" % sys.argv[2], "value %.0f" % d["value"], "steady %.0f" % (d["phases"]["steady"]["reads_per_s"] or 0), "head %.0f" % (d["phases"]["head"]["reads_per_s"] or 0), "frac %.4f" % d["roofline"]["frac"],
          "executed/useful %.4f" % (d["aux"]["kernel_stats"]["query"]["units"] / (3.0 * d["aux"]["counters"]["queries"])), "stream_inserts", t["stream_inserts"], "fallbacks", t["stream_insert_fallbacks"], "handbacks", t["stream_handbacks"],
          "relaunches", t["stream_relaunches"], "windows", t["windows"], d["aux"].get("stream_keep"), "hits", d["aux"]["counters"]["hits"], "ids", d["aux"]["counters"]["ids_inserted"])
except Exception as e:
    print("[%s]" % sys.argv[2], "FAILED", e)
PY
  grep -E "in-launch inserts|Error|error" $out/${tag}_$i.err | tail -3
done
