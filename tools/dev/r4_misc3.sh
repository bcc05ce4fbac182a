#!/bin/bash
# round 4, GPU run: decisions of long reads in registers (two / four tiles per lane), the long-tail bench, head A/B of GRP_SYNC_FR1
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
show() {
python3 - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = d["aux"]["timed"]
    st = d["phases"]["steady"].get("reads_per_s")
    print(sys.argv[2], "reads/s %.0f" % d["value"], "steady", st and round(st), "head s %.2f" % d["phases"]["head"]["seconds"],
          {k: t.get(k) for k in ("windows", "stream_inserts", "stream_insert_fallbacks", "stream_relaunches", "stream_handbacks", "batches")},
          {k: (v["launches"], round(v["ms"])) for k, v in d["aux"]["kernel_stats"].items() if v["launches"]}, "hits", d["aux"]["counters"]["hits"],
          d["config"].get("read_lengths"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
timeout 900 python3 -m pytest tests/test_reference_funcs.py tests/test_gpu_classifier.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python3 bench.py --len-sigma 0.6 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c2_long_tail.json 2> $out/r04_c2_long_tail.err; show $out/r04_c2_long_tail.json c2_long_tail
for m in 1 0 1 0; do
  if [ $m = 1 ]; then export GRP_SYNC_FR1=1; else unset GRP_SYNC_FR1; fi
  timeout 900 python3 bench.py --reads 300000 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/r04_head_fr1_$m.json 2> /dev/null; show $out/r04_head_fr1_$m.json head_fr1_$m
done
