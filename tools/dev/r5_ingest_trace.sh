#!/bin/bash
# round 5, GPU box: where the host's time of the binary's ingest + fill pass goes (GRP_TRACE_INGEST), 200 k reads = 10 GB
export TMPDIR=/tmp
gcc -O3 -fopenmp -o /tmp/fqgen tools/fqgen.c -lm && /tmp/fqgen /tmp/ingest.fq ${1:-200000} 100000000 1
CLI=goldrush_amd/bin/goldrush-path
base="-k22 -w16 -t1000 -u5 -a1 -o0.1 -h3 -j16 -P10 -d5 -x10 -s1011011110110111101101 -g100000000 -b10 --verbose -m20000 -i /tmp/ingest.fq"
for rep in 1 2 3; do
  [ $rep = 3 ] && export GRP_TRACE_NOFILL=1
  GRP_TRACE_INGEST=1 $CLI $base -p /tmp/ing_out 2>&1 | grep -E "^in |GRP_TRACE" | tr '\n' ' '; echo
done
rm -f /tmp/ingest.fq
