#!/bin/bash
# round 4, GPU box: where the CLI's ingest pass spends its time (reader threads x chunk size), 200 k reads = 10 GB of FASTQ
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from goldrush_amd import synth
t0 = time.time()
g = synth.random_genome(100_000_000, 1)
n_reads, done = 200_000, 0
with open("/tmp/ingest.fq", "wb") as fh:
    while done < n_reads:
        for rid, seq, qual in synth.make_reads(g, 5000, mean_len=25000, min_len=20000, seed=2 + done):
            fh.write(b"@r%d\n" % done + seq + b"\n+\n" + qual + b"\n")
            done += 1
print("fastq", os.path.getsize("/tmp/ingest.fq"), "bytes written in %.1f s" % (time.time() - t0))
PY
CLI=goldrush_amd/bin/goldrush-path
base="-k22 -w16 -t1000 -u5 -a1 -o0.1 -h3 -j16 -P10 -d5 -x10 -s1011011110110111101101 -g100000000 -b10 --verbose -m20000 -i /tmp/ingest.fq"
run() {
  s=$(date +%s.%N)
  env $1 $2 $CLI $base -p /tmp/ing_out > /tmp/ing.log 2>&1
  e=$(date +%s.%N)
  echo "$1 $2: wall $(python3 -c "print(round($e - $s, 2))") s; phases: $(grep -E '^in [0-9.]+$' /tmp/ing.log | tr '\n' ' ')"
}
timeout 900 python3 -m pytest tests/test_gpu_ingest.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3
run A=1 B=1
run A=1 B=2
run GRP_READ_THREADS=16 GRP_INGEST_CHUNK=134217728
run GRP_READ_THREADS=16 GRP_INGEST_CHUNK=536870912
run GRP_READ_THREADS=16 GRP_RESIDENT=off
