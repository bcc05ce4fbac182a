#!/bin/bash
# GPU box: C2 (3 M reads) against developer switches of the streaming query kernel
for s in "X=0" "GRP_STREAM_WGS_PER_CU=3" "GRP_STREAM_WGS_PER_CU=6" "GRP_STREAM_WGS_PER_CU=8" "GRP_STREAM_FR2=1"; do
  env $s python3 bench.py --no-cpu-baseline --reads 3000000 --steps 3 2>/dev/null > /tmp/ss.json
  python3 - "$s" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/ss.json") if l.startswith("{")][-1])
print(sys.argv[1].ljust(28), "reads/s", round(d["value"]), "steady reads/s", round(d["phases"]["steady"]["reads_per_s"]), "Gprobes/s", round(d["roofline"]["line_rate_Gprobes_per_s"], 2))
PY
done
