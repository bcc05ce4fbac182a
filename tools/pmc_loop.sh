#!/bin/bash
# GPU box: instruction-cache and issue counters of the commit-loop kernel (developer).
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_INSTS_(VALU|SALU|SMEM|LDS|VMEM)|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST" | head -60 > $GRAFT_REPO_ROOT/$out/r02_pmc_avail.txt
cd $GRAFT_REPO_ROOT
args="--config C1 --no-cpu-baseline --reads 100000 --steps 2 --warmup 0"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/r02_pmc_loop -o a -- python3 bench.py $args > $out/r02_pmc_loop_bench.json 2> $out/r02_pmc_loop.err
f=$(find $out/r02_pmc_loop -name "*counter_collection.csv" | head -1)
python3 - $f <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
for k, v in agg.items():
    if "commit_loop" in k or "k_query" in k or "insert" in k:
        print(k, n[k], dict(v))
PY
rm -rf $out/r02_pmc_loop
head -40 $out/r02_pmc_avail.txt
