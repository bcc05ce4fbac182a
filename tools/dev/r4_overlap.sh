#!/bin/bash
# round 4, GPU run: windows of batches that end in front of the first overlapping read (grp_window_overlap) — tests, then A/B
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_classifier.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3
show() {
python3 - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = d["aux"]["timed"]
    ps = d["aux"].get("pipeline_shaped") or {}
    print(sys.argv[2], "reads/s %.0f" % d["value"], "head s %.2f" % d["phases"]["head"]["seconds"],
          {k: t.get(k) for k in ("windows", "batches", "batches_undone", "batch_overlap_cuts", "reads_queried", "batches_fused")},
          {k: (v["launches"], round(v["ms"])) for k, v in d["aux"]["kernel_stats"].items() if v["launches"]}, "hits", d["aux"]["counters"]["hits"],
          "pipeline", ps.get("reads_per_s") and round(ps["reads_per_s"]), ps.get("batches"), ps.get("batches_undone"), ps.get("batch_overlap_cuts"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for v in "off 0.3" "8 0.3" "16 0.3" "8 0.1"; do
  set -- $v
  GRP_BATCH_OVERLAP=$1 GRP_BATCH_OVERLAP_P=$2 timeout 900 python3 bench.py --reads 300000 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/r04_ovl_head_$1_$2.json 2> /dev/null; show $out/r04_ovl_head_$1_$2.json head300k_$1_$2
  GRP_BATCH_OVERLAP=$1 GRP_BATCH_OVERLAP_P=$2 timeout 900 python3 bench.py --config C4 --reads 300000 --no-cpu-baseline > $out/r04_ovl_c4_$1_$2.json 2> /dev/null; show $out/r04_ovl_c4_$1_$2.json c4_300k_$1_$2
done
for v in "off 0.3" "8 0.3"; do
  set -- $v
  GRP_BATCH_OVERLAP=$1 GRP_BATCH_OVERLAP_P=$2 timeout 900 python3 bench.py --no-cpu-baseline > $out/r04_ovl_default_$1.json 2> /dev/null; show $out/r04_ovl_default_$1.json default_$1
done
