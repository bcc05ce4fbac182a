#!/bin/bash
# GPU box: the default bench command under rocprofv3 --kernel-trace --stats, and the timed region's
# k_query dispatches from the trace next to the bench line's HIP-event average.
#   tools/rocprof_round.sh <tag>
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
# kernel trace of the default command (no CPU baseline leg: it only adds host time)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 bench.py --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
find $out/${tag}_prof -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
# the bench line's roofline is over the throughput forms of k_query in the timed region: the same
# dispatches from the per-dispatch trace (streaming + large windows; the warm-up engine's launches
# come first and are cut off by taking the LAST <launches> of them)
python3 - $out/${tag}_bench_under_rocprof.json $(find $out/${tag}_prof -name "*kernel_trace.csv" | head -1) $out/${tag}_kernel_trace_timed_region.json <<'PY'
import csv, json, sys
b = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
n = int(b["roofline"]["launches"])
def throughput_form(name):  # k_query<H, FR, WT, ST, VER>: WT = 0 (streaming windows, large windows, batch re-queries); WT = 16 = latency windows
    if "k_query<" not in name:
        return False
    args = name[name.index("k_query<") + 8:].split(">")[0].split(", ")
    return len(args) >= 3 and args[2] == "0"
rows = [r for r in csv.DictReader(open(sys.argv[2])) if throughput_form(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
json.dump({"note": "rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline`: the last <launches> dispatches of the throughput forms of k_query = the timed region of the bench line",
           "launches": n, "avg_launch_ms_rocprof": sum(dur) / len(dur) / 1e6, "avg_launch_ms_bench_hip_events": b["roofline"]["avg_launch_ms"],
           "total_ms_rocprof": sum(dur) / 1e6, "query_kernel_ms_bench": b["aux"]["query_kernel_s"] * 1e3}, open(sys.argv[3], "w"), indent=1)
print(open(sys.argv[3]).read())
PY
rm -rf $out/${tag}_prof
