#!/bin/bash
# round 5, GPU box: per-kernel time of the insert-heavy regime (C2's reads through silver mode, 5 paths) under rocprofv3 --kernel-trace --stats
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/silver_prof -o sp -- python3 bench.py --reads 1300000 --steps 2 --silver 5 --no-cpu-baseline ${EXTRA} > $out/r05_silver_prof.json 2> /dev/null
f=$(find $out/silver_prof -name "*kernel_stats.csv" | head -1)
cp $f $out/${1:-r05}_silver_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-64s %7s calls %9.1f ms  avg %8.3f ms  %4.1f %%" % (r["Name"].split("(")[0][-64:], r["Calls"], int(r["TotalDurationNs"]) / 1e6, int(r["TotalDurationNs"]) / 1e6 / int(r["Calls"]), 100.0 * int(r["TotalDurationNs"]) / tot))
PY
rm -rf $out/silver_prof
