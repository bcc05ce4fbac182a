#!/bin/bash
# round 4, GPU run: the last host change (overlap blocks that slide with the windows) on the head and the C4 geometry, then the PMC
# passes and the driver's command on exactly this build (bench.py applies a PMC summary only to the sources it was taken on)
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
tag=${1:-r04_v7}
show() {
python3 - $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["aux"]["timed"]
ps = d["aux"].get("pipeline_shaped") or {}
print(sys.argv[2], "reads/s %.0f" % d["value"], "head s %.2f" % d["phases"]["head"]["seconds"], {k: t.get(k) for k in ("batches", "batches_undone", "batch_overlap_cuts", "reads_queried")},
      "frac %.3f" % d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "pipeline", ps.get("reads_per_s") and round(ps["reads_per_s"]), (d["aux"].get("oracle_check") or {}).get("identical"))
PY
}
timeout 900 python3 bench.py --reads 300000 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_head_300k.json 2> /dev/null; show $out/${tag}_head_300k.json head300k
timeout 900 python3 bench.py --config C4 --reads 4000000 --no-cpu-baseline > $out/${tag}_bench_c4_4M.json 2> /dev/null; show $out/${tag}_bench_c4_4M.json c4_4M
tools/pmc_round.sh ${tag} > $out/${tag}_pmc.log 2>&1
cp $out/${tag}_pmc_summary.json profiles/ 2>/dev/null
python3 bench.py > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err; show $out/${tag}_bench_default_flags.json default
tools/rocprof_round.sh ${tag} > $out/${tag}_rocprof_round.log 2>&1; tail -9 $out/${tag}_rocprof_round.log | head -8
