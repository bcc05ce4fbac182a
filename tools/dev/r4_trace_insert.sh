#!/bin/bash
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
GRP_TRACE_ABORT=1 timeout 900 python3 bench.py --reads 2500000 --steps 5 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/r04_trace_abort.json 2> $out/r04_trace_abort.err
grep -E "in-launch inserts|stream rounds" $out/r04_trace_abort.err | tail -6
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_trace_abort.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["phases"]["steady"], d["aux"]["timed"])
PY
