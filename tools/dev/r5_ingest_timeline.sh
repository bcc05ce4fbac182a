#!/bin/bash
# round 5, GPU box: the timeline of the binary's ingest + fill pass (100 k reads, 5 GB): kernels and copies of three chunks in the middle
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
gcc -O3 -fopenmp -o /tmp/fqgen tools/fqgen.c -lm && /tmp/fqgen /tmp/ingest.fq 100000 100000000 1
CLI=goldrush_amd/bin/goldrush-path
base="-k22 -w16 -t1000 -u5 -a1 -o0.1 -h3 -j16 -P10 -d5 -x10 -s1011011110110111101101 -g100000000 -b10 --verbose -m20000 -i /tmp/ingest.fq"
$CLI $base -p /tmp/ing_out 2>&1 | grep -E "^in " | tr '\n' ' '; echo
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/ingest_tl -o ing -- $CLI $base -p /tmp/ing_out > /tmp/ing.log 2>&1
grep -E "^in " /tmp/ing.log | tr '\n' ' '; echo
k=$(find $out/ingest_tl -name "*kernel_trace.csv" | head -1)
m=$(find $out/ingest_tl -name "*memory_copy_trace.csv" | head -1)
python3 - $k $m <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].split("<")[0][-28:], r.get("Stream_Id", r.get("Queue_Id", ""))))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r["Direction"][:14], r.get("Stream_Id", "")))
ev.sort()
fills = [e for e in ev if "k_fill" in e[2]]
if len(fills) > 12:
    t0 = fills[8][0] - 2_000_000
    t1 = fills[11][1] + 2_000_000
    print("events between the 9th and the 12th fill (ms from the window's start; only those >= 0.05 ms or fills / big copies):")
    for s, e, n, q in ev:
        if s >= t0 and s <= t1 and ((e - s) >= 20_000 or n.startswith("C ")):
            print("%8.2f .. %8.2f  %6.2f ms  stream %s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
PY
rm -rf $out/ingest_tl /tmp/ingest.fq
