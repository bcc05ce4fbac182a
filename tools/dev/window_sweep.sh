#!/bin/bash
# GPU box: steady-state rate of C2 (5 M reads) against the size of the streaming windows
for w in 2048 4096 16384; do
  python3 bench.py --no-cpu-baseline --max-window $w --reads 5000000 --steps 5 2>/dev/null > /tmp/ws.json
  python3 - $w <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/ws.json") if l.startswith("{")][-1])
print("max_window", sys.argv[1], "reads/s", round(d["value"]), "head s", round(d["phases"]["head"]["seconds"], 2), "steady reads/s", round(d["phases"]["steady"]["reads_per_s"]))
PY
done
