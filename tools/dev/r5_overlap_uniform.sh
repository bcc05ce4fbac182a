#!/bin/bash
# round 5, GPU run: the insert-heavy regime (silver mode, 5 paths) on the UNIFORM genome by the overlap threshold
out=gpurun_out
mkdir -p $out
for ov in ${OVS:-8 32 64 256}; do
  GRP_BATCH_OVERLAP=$ov timeout 600 python3 bench.py --reads 1300000 --steps 2 --silver 5 --no-cpu-baseline > $out/r05_uni_ov_$ov.json 2> /dev/null
  python3 - $out/r05_uni_ov_$ov.json $ov <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["aux"]["timed"]
ks = {n: round(v["ms"]) for n, v in d["aux"]["kernel_stats"].items() if v["launches"]}
print("overlap", sys.argv[2], "reads/s %.0f" % d["value"], "reads", d["config"]["reads_timed"], {k: t[k] for k in ("batches", "batches_undone", "batch_reads", "reads_queried", "inserts", "batch_overlap_cuts")}, ks)
PY
done
