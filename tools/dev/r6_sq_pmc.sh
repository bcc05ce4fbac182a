#!/bin/bash
# GPU box, round 6: SQ counters of the query kernels on the C1 stream (the streaming kernel with its fingerprint stores)
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/r06_sq -o sq -- python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped --steps 4 --warmup 0 > $out/r06_sq_bench.json 2> $out/r06_sq.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r06_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][-60:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[k].add(row["Dispatch_Id"])
for k, c in sorted(acc.items(), key=lambda t: -t[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    w = c.get("SQ_WAVE_CYCLES", 1)
    print("%-62s disp %6d  wave_cycles %.3e  wait_any %.2f  wait_inst %.2f  active %.2f  valu %.2f  lds %.2f  wait_lds %.3f  insts_valu %.3e" % (
        k, len(n[k]), w, c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_VALU", 0) / w,
        c.get("SQ_ACTIVE_INST_LDS", 0) / w, c.get("SQ_WAIT_INST_LDS", 0) / w, c.get("SQ_INSTS_VALU", 0)))
PY
rm -rf $out/r06_sq
