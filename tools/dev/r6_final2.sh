#!/bin/bash
# round 6, GPU box, second part of the evidence on the final sources: the GPU suite, the 1-rank / 2-rank pairs the scale model
# counts from, the PMC passes (bench.py applies a summary only to the sources it was taken on) and the driver's command.
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
tag=${1:-r06_final}
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $out/${tag}_gpu_tests.txt; tail -3 $out/${tag}_gpu_tests.txt
tools/dev/rank_truth_check.sh C1 300000 ${tag} > $out/${tag}_truth_ranks.txt 2>&1; tail -1 $out/${tag}_truth_ranks.txt
tools/dev/rank_truth_check.sh C2 ${2:-1000000} ${tag} >> $out/${tag}_truth_ranks.txt 2>&1; tail -1 $out/${tag}_truth_ranks.txt
tools/pmc_round.sh ${tag} > $out/${tag}_pmc.log 2>&1
cp $out/${tag}_pmc_summary.json profiles/ 2>/dev/null
timeout 900 python3 bench.py > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err
python3 - $out/${tag}_bench_default_flags.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default: reads/s %.0f" % d["value"], "frac %.3f" % d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "fill traffic", (d.get("roofline_fill") or {}).get("traffic"), "pipeline", (d["aux"].get("pipeline_shaped") or {}).get("reads_per_s"),
      "oracle", (d["aux"].get("oracle_check") or {}).get("identical"), "fit", d["phases"]["steady"].get("fit"))
PY
python3 tools/scale_model.py $out/${tag}_bench_default_flags.json --ranks $out/${tag}_ranks_C2_one.json $out/${tag}_ranks_C2_two.json > $out/${tag}_scale_model.txt 2>&1; tail -6 $out/${tag}_scale_model.txt
