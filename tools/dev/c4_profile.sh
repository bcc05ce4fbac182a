#!/bin/bash
# GPU box: per-kernel stats of the first reads of the C4 geometry (h = 5, silver mode)
out=gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4_prof -o c4 -- python3 bench.py --config C4 --reads ${1:-400000} --steps 2 --warmup 0 --no-cpu-baseline > $out/c4_head.json 2> $out/c4_head.err
find $out/c4_prof -name "*kernel_stats.csv" -exec cp {} $out/c4_head_kernel_stats.csv \;
rm -rf $out/c4_prof
python3 - <<'PY'
import csv, re, json
for r in csv.DictReader(open("gpurun_out/c4_head_kernel_stats.csv")):
    n = r["Name"]; m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    if int(r["TotalDurationNs"]) > 2e7:
        print((m.group(1) if m else n)[:40].ljust(40), r["Calls"].rjust(6), str(round(int(r["TotalDurationNs"]) / 1e6, 1)).rjust(8), "ms avg", round(float(r["AverageNs"]) / 1e3, 1), "us")
d = json.loads([l for l in open("gpurun_out/c4_head.json") if l.startswith("{")][-1])
print(d["value"], d["aux"]["timed"])
PY
