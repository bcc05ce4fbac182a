#!/bin/bash
# round 4, GPU box: kernel time of the CLI's passes (100 k reads, 5 GB of FASTQ) under rocprofv3 --kernel-trace --stats
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from goldrush_amd import synth
g = synth.random_genome(100_000_000, 1)
n_reads, done = 100_000, 0
with open("/tmp/ingest.fq", "wb") as fh:
    while done < n_reads:
        for rid, seq, qual in synth.make_reads(g, 5000, mean_len=25000, min_len=20000, seed=2 + done):
            fh.write(b"@r%d\n" % done + seq + b"\n+\n" + qual + b"\n")
            done += 1
print("fastq", os.path.getsize("/tmp/ingest.fq"))
PY
CLI=goldrush_amd/bin/goldrush-path
base="-k22 -w16 -t1000 -u5 -a1 -o0.1 -h3 -j16 -P10 -d5 -x10 -s1011011110110111101101 -g100000000 -b10 --verbose -m20000 -i /tmp/ingest.fq"
$CLI $base -p /tmp/ing_out 2>&1 | grep -E "^in " | tr '\n' ' '; echo
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/ingest_prof -o ing -- $CLI $base -p /tmp/ing_out > /tmp/ing.log 2>&1
grep -E "^in " /tmp/ing.log | tr '\n' ' '; echo
f=$(find $out/ingest_prof -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s %6s calls %8.1f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], int(r["TotalDurationNs"]) / 1e6))
PY
m=$(find $out/ingest_prof -name "*memory_copy_trace.csv" | head -1)
python3 - $m <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(len(rows), "copies, total %.1f ms" % sum(d))
for r, x in sorted(zip(rows, d), key=lambda t: -t[1])[:24]:
    print(r["Direction"], "stream", r["Stream_Id"], "start %.1f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6), "%.2f ms" % x)
PY
rm -rf $out/ingest_prof
