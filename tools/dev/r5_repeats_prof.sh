#!/bin/bash
# round 5, GPU box: per-kernel time of the golden pass on the repeat-rich genome (bench.py --repeat-frac 0.4) under rocprofv3 --kernel-trace --stats
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rep_prof -o rp -- python3 bench.py --repeat-frac 0.4 --no-cpu-baseline --no-pipeline-shaped ${EXTRA} > $out/r05_repeats_prof.json 2> /dev/null
f=$(find $out/rep_prof -name "*kernel_stats.csv" | head -1)
cp $f $out/${1:-r05}_repeats_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-64s %7s calls %9.1f ms  avg %8.3f ms  max %8.3f ms  %4.1f %%" % (r["Name"].split("(")[0][-64:], r["Calls"], int(r["TotalDurationNs"]) / 1e6, int(r["TotalDurationNs"]) / 1e6 / int(r["Calls"]), int(r["MaxNs"]) / 1e6, 100.0 * int(r["TotalDurationNs"]) / tot))
PY
rm -rf $out/rep_prof
