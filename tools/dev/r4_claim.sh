#!/bin/bash
# round 4, GPU run: the claim of the collect pass folded into the count word — batch / verify / classifier suites, then the head, C4 and the default bench
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_classifier.py tests/test_gpu_stream_insert.py tests/test_gpu_parity.py tests/test_reference_mini.py tests/test_reference_funcs.py -m gpu -x -q > $out/r04_claim_tests.log 2>&1; tail -5 $out/r04_claim_tests.log
show() {
python3 - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = d["aux"]["timed"]
    st = d["phases"]["steady"].get("reads_per_s")
    print(sys.argv[2], "reads/s %.0f" % d["value"], "steady", st and round(st), "head s %.2f" % d["phases"]["head"]["seconds"],
          {k: t.get(k) for k in ("windows", "batches", "batches_undone")},
          {k: (v["launches"], round(v["ms"])) for k, v in d["aux"]["kernel_stats"].items() if v["launches"]}, "hits", d["aux"]["counters"]["hits"], (d["aux"].get("pipeline_shaped") or {}).get("reads_per_s"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
timeout 900 python3 bench.py --reads 300000 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/r04_claim_head.json 2> $out/r04_claim_head.err; show $out/r04_claim_head.json head300k
timeout 900 python3 bench.py --config C4 --reads 300000 --no-cpu-baseline > $out/r04_claim_c4.json 2> $out/r04_claim_c4.err; show $out/r04_claim_c4.json c4_300k
timeout 900 python3 bench.py --no-cpu-baseline > $out/r04_claim_default.json 2> $out/r04_claim_default.err; show $out/r04_claim_default.json default
