#!/bin/bash
# GPU box (round 3): persistent workgroups per CU / window length against the cost of an in-launch insert (C2)
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
run() { # name, env..., -- bench args
  name=$1; shift
  env "$@" GRP_TRACE_ABORT=1 timeout 900 python3 bench.py --config C2 --no-cpu-baseline $EXTRA > $out/r03_sweep_$name.json 2> $out/r03_sweep_$name.err
  python3 - $out/r03_sweep_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ph = d["phases"]
    print(sys.argv[2], "value %.0f" % d["value"], "head %.2f s" % ph["head"]["seconds"], "steady %.0f" % ph["steady"]["reads_per_s"], "windows", d["aux"]["timed"]["windows"], "queried", d["aux"]["timed"]["reads_queried"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  grep "in-launch" $out/r03_sweep_$name.err | tail -1
}
run wg4 A=1
run wg3 GRP_STREAM_WGS_PER_CU=3
run wg2 GRP_STREAM_WGS_PER_CU=2
EXTRA="--max-window 65536" run wg4_w64k A=1
EXTRA="--max-window 65536" run wg3_w64k GRP_STREAM_WGS_PER_CU=3
