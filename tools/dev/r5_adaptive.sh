#!/bin/bash
# round 5, GPU run: the adaptive overlap threshold on both genomes (silver mode, 5 paths), then the tests that cover decisions
out=gpurun_out
python -m pytest tests/test_reference_funcs.py tests/test_gpu_classifier.py tests/test_gpu_stream_insert.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -2 | head -1
for g in uni rep; do
  extra=""; [ $g = rep ] && extra="--repeat-frac 0.4"
  timeout 600 python3 bench.py --reads 1500000 --steps 2 --silver 5 $extra --no-cpu-baseline > $out/r05_adapt_$g.json 2> /dev/null
  python3 - $out/r05_adapt_$g.json $g <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["aux"]["timed"]
ks = {n: round(v["ms"]) for n, v in d["aux"]["kernel_stats"].items() if v["launches"]}
print(sys.argv[2], "adaptive: reads/s %.0f" % d["value"], "reads", d["config"]["reads_timed"], {k: t[k] for k in ("batches", "batches_undone", "batch_reads", "reads_queried", "inserts", "batch_overlap_cuts")}, ks)
PY
done
