#!/bin/bash
# round 4, GPU run: wide spans, claim-loop guard, decisions of very long reads (LDS state), long-tail bench
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_reference_funcs.py tests/test_gpu_parity.py "tests/test_gpu_batch.py::test_claim_loops_keep_every_touch" tests/test_gpu_classifier.py tests/test_gpu_stream_insert.py -m gpu -x -q > $out/r04_misc4_tests.log 2>&1; tail -15 $out/r04_misc4_tests.log
timeout 900 python3 bench.py --len-sigma 0.6 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c2_long_tail.json 2> $out/r04_c2_long_tail.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_c2_long_tail.json").read().strip().splitlines()[-1])
t = d["aux"]["timed"]
print("long_tail reads/s %.0f" % d["value"], "steady %.0f" % d["phases"]["steady"]["reads_per_s"], "head s %.2f" % d["phases"]["head"]["seconds"],
      {k: t.get(k) for k in ("windows", "stream_inserts", "stream_handbacks", "batches")}, {k: (v["launches"], round(v["ms"])) for k, v in d["aux"]["kernel_stats"].items() if v["launches"]}, d["config"].get("read_lengths"))
PY
python3 bench.py --no-cpu-baseline > $out/r04_v4_bench_default.json 2> $out/r04_v4_bench_default.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_v4_bench_default.json").read().strip().splitlines()[-1])
t = d["aux"]["timed"]
print("default reads/s %.0f" % d["value"], "steady %.0f" % d["phases"]["steady"]["reads_per_s"], "head s %.2f" % d["phases"]["head"]["seconds"], "frac %.3f" % d["roofline"]["frac"],
      {k: (v["launches"], round(v["ms"])) for k, v in d["aux"]["kernel_stats"].items() if v["launches"]}, d["aux"].get("pipeline_shaped", {}).get("reads_per_s"))
PY
