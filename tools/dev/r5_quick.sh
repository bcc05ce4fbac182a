#!/bin/bash
# round 5, GPU run: batch / classifier suites, then the default bench without the CPU leg; one summary line each
#   tools/dev/r5_quick.sh <tag> [pytest files...]
tag=${1:-r05_x}; shift
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
tests=${@:-tests/test_gpu_batch.py tests/test_gpu_classifier.py}
timeout 1500 python3 -m pytest $tests -m gpu -x -q > $out/${tag}_tests.log 2>&1; tail -3 $out/${tag}_tests.log
timeout 900 python3 bench.py --no-cpu-baseline > $out/${tag}_bench_nocpu.json 2> $out/${tag}_bench_nocpu.err
python3 - $out/${tag}_bench_nocpu.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = lambda k: {n: (v["launches"], round(v["ms"])) for n, v in k.items() if v["launches"]}
print("reads/s %.0f frac %.3f head %.2fs steady %.0f" % (d["value"], d["roofline"]["frac"], d["phases"]["head"]["seconds"], d["phases"]["steady"]["reads_per_s"]), ks(d["aux"]["kernel_stats"]))
bi = d["aux"]["kernel_stats"]["batch_insert"]; print("collect+apply G rec/s %.2f" % (bi["units"] / bi["ms"] / 1e6))
ps = d["aux"].get("pipeline_shaped")
if ps:
    print("pipeline_shaped %.0f reads/s %.2fs" % (ps["reads_per_s"], ps["seconds"]), ks(ps["kernel_stats"]))
    bi = ps["kernel_stats"]["batch_insert"]; print("collect+apply G rec/s %.2f" % (bi["units"] / bi["ms"] / 1e6))
print("counters", d["aux"]["counters"])
PY
