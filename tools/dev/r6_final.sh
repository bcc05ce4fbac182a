#!/bin/bash
# round 6, GPU box: the evidence of the final build — the three PMC passes and the gather ceiling first (bench.py applies a
# PMC summary only to the sources it was taken on), then the driver's command, the same command under rocprofv3, the C1
# full stream, the silver pass's kernel stats and the C4 geometry.      tools/dev/r6_final.sh <tag>
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
tag=${1:-r06_final}
show() {
python3 - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    t = d["aux"]["timed"]
    ps = d["aux"].get("pipeline_shaped") or {}
    fit = (d["phases"]["steady"].get("fit") or {})
    print(sys.argv[2], "reads/s %.0f" % d["value"], "head %.2f s" % d["phases"]["head"]["seconds"], "steady %.0f" % (d["phases"]["steady"]["reads_per_s"] or 0), "per insert %.1f us" % (1e6 * fit.get("s_per_insert", 0)),
          "frac %.3f" % d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "fill frac", (d.get("roofline_fill") or {}).get("frac"), "pipeline", ps.get("reads_per_s") and round(ps["reads_per_s"]),
          "oracle", (d["aux"].get("oracle_check") or {}).get("identical"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "fallbacks", t.get("stream_insert_fallbacks"), d["aux"].get("stream_keep"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
timeout 600 python3 tools/gather_ceiling.py $out/${tag}_gather_ceiling.json > /dev/null 2>&1
cp $out/${tag}_gather_ceiling.json profiles/ 2>/dev/null
tools/pmc_round.sh ${tag} > $out/${tag}_pmc.log 2>&1
cp $out/${tag}_pmc_summary.json profiles/ 2>/dev/null
timeout 900 python3 bench.py > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err; show $out/${tag}_bench_default_flags.json default
tools/rocprof_round.sh ${tag} > $out/${tag}_rocprof_round.log 2>&1; show $out/${tag}_bench_under_rocprof.json under_rocprof; cat $out/${tag}_kernel_trace_timed_region.json
timeout 600 python3 bench.py --config C1 --no-cpu-baseline > $out/${tag}_bench_c1_full_stream.json 2> /dev/null; show $out/${tag}_bench_c1_full_stream.json c1
timeout 900 python3 bench.py --config C4 --reads 4000000 --no-cpu-baseline > $out/${tag}_bench_c4_4M.json 2> /dev/null; show $out/${tag}_bench_c4_4M.json c4_4M
EXTRA="" tools/dev/r5_silver_prof.sh ${tag} > $out/${tag}_silver_prof.txt 2>&1; tail -16 $out/${tag}_silver_prof.txt
