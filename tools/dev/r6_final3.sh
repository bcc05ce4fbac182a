#!/bin/bash
# round 6, GPU box: the PMC passes and the driver's command on the final sources, and a short soak of the default workload
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
tag=${1:-r06_final}
tools/pmc_round.sh ${tag} > $out/${tag}_pmc.log 2>&1
cp $out/${tag}_pmc_summary.json profiles/ 2>/dev/null
timeout 900 python3 bench.py > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err
for i in 1 2 3; do
  timeout 600 python3 bench.py --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_soak_$i.json 2> /dev/null
done
python3 - $out/${tag}_bench_default_flags.json $out/${tag}_soak_1.json $out/${tag}_soak_2.json $out/${tag}_soak_3.json <<'PY'
import json, sys
ref = None
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e)
        continue
    c = (d["aux"]["counters"], d["aux"]["pop"])
    ref = ref or c
    t = d["aux"]["timed"]
    print(f.split("/")[-1], "reads/s %.0f" % d["value"], "steady %.0f" % d["phases"]["steady"]["reads_per_s"], "frac %.3f" % d["roofline"]["frac"], "traffic", d["roofline"]["traffic"] and round(d["roofline"]["traffic"] / 1e9, 2),
          "per insert us %.0f" % (1e6 * (d["phases"]["steady"].get("fit") or {}).get("s_per_insert", 0)), "in-launch inserts", t["stream_inserts"], "fallbacks", t["stream_insert_fallbacks"], "handbacks", t["stream_handbacks"],
          "counters", "identical" if c == ref else "DIFFERENT", "pipeline", (d["aux"].get("pipeline_shaped") or {}).get("reads_per_s"), "oracle", (d["aux"].get("oracle_check") or {}).get("identical"))
PY
