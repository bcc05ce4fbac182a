#!/bin/bash
# Run on the GPU box (gpurun): refresh the judged evidence for the current build.
#   tools/profile_round.sh <tag>      e.g. r01_v8
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err
python bench.py --steps 112 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_full_stream.json 2> /dev/null
# kernel trace of the same default command (no CPU baseline leg: it only adds host time)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 bench.py --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
find $out/${tag}_prof -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
# the stats above cover warm-up + timed steps; the bench line only the timed ones: average the
# LAST <launches> dispatches of the query kernels from the per-dispatch trace as well
python3 - $out/${tag}_bench_under_rocprof.json $(find $out/${tag}_prof -name "*kernel_trace.csv" | head -1) $out/${tag}_kernel_trace_timed_region.json <<'PY'
import csv, json, sys
b = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
n = int(b["roofline"]["launches"])
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_query" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
json.dump({"note": "rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline`: the last <launches> k_query dispatches = the timed region of the bench line",
           "launches": n, "avg_launch_ms_rocprof": sum(dur) / len(dur) / 1e6, "avg_launch_ms_bench_hip_events": b["roofline"]["avg_launch_ms"],
           "total_ms_rocprof": sum(dur) / 1e6, "query_kernel_ms_bench": b["aux"]["query_kernel_s"] * 1e3}, open(sys.argv[3], "w"), indent=1)
print(open(sys.argv[3]).read())
PY
rm -rf $out/${tag}_prof
# two ranks on the one GPU (gloo): plumbing of the N > 1 path
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 12 --warmup 4 --no-cpu-baseline --backend gloo --share-gpu --reads 400000 --verify-ranks > $out/${tag}_bench_2ranks_one_gpu_gloo_plumbing.json 2> $out/${tag}_2ranks.err
grep verify-ranks $out/${tag}_2ranks.err; tail -2 $out/${tag}_2ranks.err
cat $out/${tag}_bench_default_flags.json $out/${tag}_bench_full_stream.json $out/${tag}_bench_under_rocprof.json $out/${tag}_bench_2ranks_one_gpu_gloo_plumbing.json
head -8 $out/${tag}_kernel_stats.csv
