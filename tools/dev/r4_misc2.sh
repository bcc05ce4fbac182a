#!/bin/bash
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
show() {
python3 - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = d["aux"]["timed"]
    print(sys.argv[2], "reads/s %.0f" % d["value"], "steady %.0f" % d["phases"]["steady"]["reads_per_s"], "head s %.2f" % d["phases"]["head"]["seconds"],
          {k: t.get(k) for k in ("windows", "stream_inserts", "stream_insert_fallbacks", "stream_relaunches", "stream_handbacks", "batches")},
          {k: (v["launches"], round(v["ms"])) for k, v in d["aux"]["kernel_stats"].items() if v["launches"]}, "Gprobes/s %.1f" % d["roofline"]["line_rate_Gprobes_per_s"], "hits", d["aux"]["counters"]["hits"],
          d["config"].get("read_lengths"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for m in on off; do
  GRP_STREAM_COOP=$m timeout 600 python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c1_coop_$m.json 2> $out/r04_c1_coop_$m.err; show $out/r04_c1_coop_$m.json c1_coop_$m; tail -1 $out/r04_c1_coop_$m.err | cut -c1-300
done
timeout 900 python3 bench.py --len-sigma 0.6 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c2_long_tail.json 2> $out/r04_c2_long_tail.err; show $out/r04_c2_long_tail.json c2_long_tail
for m in 1 0; do
  if [ $m = 1 ]; then export GRP_SYNC_FR1=1; else unset GRP_SYNC_FR1; fi
  timeout 900 python3 bench.py --reads 300000 --steps 2 --warmup 0 --no-cpu-baseline --no-pipeline-shaped > $out/r04_head_fr1_$m.json 2> /dev/null; show $out/r04_head_fr1_$m.json head_fr1_$m
done
unset GRP_SYNC_FR1
python3 bench.py > $out/r04_v3_bench_default.json 2> $out/r04_v3_bench_default.err; show $out/r04_v3_bench_default.json c2_default
python3 -c "
import json
d=json.loads(open('$out/r04_v3_bench_default.json').read().strip().splitlines()[-1])
print(d['aux'].get('oracle_check'), d.get('cpu_baseline',{}).get('value'), d['aux'].get('pipeline_shaped',{}).get('reads_per_s'), d['roofline']['frac'])"
