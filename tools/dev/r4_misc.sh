#!/bin/bash
# GPU box (round 4): cooperative launch on / off on the C1 stream; the C2 stream with a realistic length tail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for m in on off on off; do
  GRP_STREAM_COOP=$m timeout 600 python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c1_coop_$m.json 2> /dev/null
  python3 - $out/r04_c1_coop_$m.json coop_$m <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "reads/s %.0f" % d["value"], "steady %.0f" % d["phases"]["steady"]["reads_per_s"], "head s %.2f" % d["phases"]["head"]["seconds"], "stream_inserts", d["aux"]["timed"]["stream_inserts"], "hits", d["aux"]["counters"]["hits"])
PY
done
timeout 900 python3 bench.py --len-sigma 0.6 --no-cpu-baseline --no-pipeline-shaped > $out/r04_c2_long_tail.json 2> $out/r04_c2_long_tail.err
python3 - $out/r04_c2_long_tail.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("long tail", "reads/s %.0f" % d["value"], d["config"]["read_lengths"], "steady %.0f" % d["phases"]["steady"]["reads_per_s"], "head s %.2f" % d["phases"]["head"]["seconds"], "frac %.3f" % d["roofline"]["frac"], "Gprobes/s %.1f" % d["roofline"]["line_rate_Gprobes_per_s"])
PY
