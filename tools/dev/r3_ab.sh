#!/bin/bash
# GPU box (round 3): in-launch inserts of streaming windows, A/B on the same box.
#   tools/dev/r3_ab.sh <tag>
tag=${1:-r03}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for cfg in C1 C2; do
  timeout 900 python3 bench.py --config $cfg --no-cpu-baseline > $out/${tag}_ab_${cfg}_resume.json 2> $out/${tag}_ab_${cfg}_resume.err
  GRP_STREAM_RESUME=off timeout 900 python3 bench.py --config $cfg --no-cpu-baseline > $out/${tag}_ab_${cfg}_classic.json 2> $out/${tag}_ab_${cfg}_classic.err
  for m in resume classic; do
    python3 - $out/${tag}_ab_${cfg}_$m.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ph = d.get("phases", {})
    print(sys.argv[1], "value %.0f" % d["value"], "head", ph.get("head", {}).get("seconds"), "steady", ph.get("steady", {}).get("reads_per_s"), "frac %.3f" % d["roofline"]["frac"], "line_rate_frac", d["roofline"].get("line_rate_frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
