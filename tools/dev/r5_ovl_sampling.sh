#!/bin/bash
# round 5, GPU box: the overlap pre-pass with half / a quarter of its samples (GRP_BATCH_OVERLAP_SHIFT) and the threshold scaled with them,
# silver pass over C2's reads (uniform genome) and over the repeat-rich genome
out=gpurun_out
mkdir -p $out
run() { # name, extra bench args, env...
  local name=$1; local args=$2; shift 2
  env "$@" python3 bench.py $args --steps 2 --silver 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r05_ovls_$name.json
  python3 - $out/r05_ovls_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d["aux"]["timed"]
print("%-22s %7.0f reads/s  batches %5d undone %4d cuts %5d queried %8d" % (sys.argv[2], d["value"], t["batches"], t["batches_undone"], t["batch_overlap_cuts"], t["reads_queried"]))
PY
}
run uni_16_T8   "--reads 1300000" X=1
run uni_32_T4   "--reads 1300000" GRP_BATCH_OVERLAP_SHIFT=59 GRP_BATCH_OVERLAP=4+
run uni_64_T2   "--reads 1300000" GRP_BATCH_OVERLAP_SHIFT=58 GRP_BATCH_OVERLAP=2+
run uni_32_T5   "--reads 1300000" GRP_BATCH_OVERLAP_SHIFT=59 GRP_BATCH_OVERLAP=5+
run rep_16_T8   "--reads 1600000 --repeat-frac 0.4" X=1
run rep_32_T4   "--reads 1600000 --repeat-frac 0.4" GRP_BATCH_OVERLAP_SHIFT=59 GRP_BATCH_OVERLAP=4+
run rep_64_T2   "--reads 1600000 --repeat-frac 0.4" GRP_BATCH_OVERLAP_SHIFT=58 GRP_BATCH_OVERLAP=2+
