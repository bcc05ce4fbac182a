#!/bin/bash
# GPU box: the counters of a 1-rank run against those of a 2-rank run (two ranks on the one GPU)
# of the same stream: aux.counters and pop must be identical.   tools/dev/rank_truth_check.sh [config] [reads]
cfg=${1:-C1}
reads=${2:-300000}
python3 bench.py --config $cfg --reads $reads --steps 6 --no-cpu-baseline 2>/dev/null > /tmp/one.json
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config $cfg --steps 6 --no-cpu-baseline --backend gloo --share-gpu --reads $reads --verify-ranks 2>/dev/null > /tmp/two.json
mkdir -p gpurun_out; cp /tmp/one.json gpurun_out/${3:-r05}_ranks_${cfg}_one.json; cp /tmp/two.json gpurun_out/${3:-r05}_ranks_${cfg}_two.json
python3 - <<'PY'
import json
a = json.loads([l for l in open("/tmp/one.json") if l.startswith("{")][-1])
b = json.loads([l for l in open("/tmp/two.json") if l.startswith("{")][-1])
print("1 rank :", a["aux"]["pop"], a["aux"]["counters"])
print("2 ranks:", b["aux"]["pop"], b["aux"]["counters"])
keys = ("windows", "reads_queried", "batches", "stream_inserts", "stream_insert_fallbacks", "stream_relaunches")
print("1 rank  timed:", {k: a["aux"]["timed"][k] for k in keys}, "query units", a["aux"]["kernel_stats"]["query"]["units"])
print("2 ranks timed (rank 0):", {k: b["aux"]["timed"][k] for k in keys}, "query units (rank 0)", b["aux"]["kernel_stats"]["query"]["units"], "comm", b["aux"].get("comm"))
print("IDENTICAL" if (a["aux"]["pop"], a["aux"]["counters"]) == (b["aux"]["pop"], b["aux"]["counters"]) else "DIFFERENT")
PY
