#!/bin/bash
# GPU box (round 4): what slows the C1 stream — persistent workgroups per CU, cooperative launch
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
show() {
python3 - $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "reads/s %.0f" % d["value"], "steady %.0f" % d["phases"]["steady"]["reads_per_s"], "head s %.2f" % d["phases"]["head"]["seconds"], "stream_inserts", d["aux"]["timed"]["stream_inserts"],
          "query ms %.0f" % d["aux"]["kernel_stats"]["query"]["ms"], "Gprobes/s %.1f" % d["roofline"]["line_rate_Gprobes_per_s"], "hits", d["aux"]["counters"]["hits"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
GRP_STREAM_COOP=on timeout 600 python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c1_a.json 2> $out/r04_c1_a.err; show $out/r04_c1_a.json coop_on; tail -3 $out/r04_c1_a.err
GRP_STREAM_COOP=off GRP_STREAM_WGS_PER_CU=3 timeout 600 python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c1_b.json 2> $out/r04_c1_b.err; show $out/r04_c1_b.json coop_off_wgs3
GRP_STREAM_COOP=off GRP_STREAM_WGS_PER_CU=4 timeout 600 python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c1_c.json 2> $out/r04_c1_c.err; show $out/r04_c1_c.json coop_off_wgs4
GRP_STREAM_COOP=off GRP_STREAM_WGS_PER_CU=2 timeout 600 python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c1_d.json 2> $out/r04_c1_d.err; show $out/r04_c1_d.json coop_off_wgs2
