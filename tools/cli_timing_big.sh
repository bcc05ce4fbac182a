# developer tool: goldrush-path CLI phases on ~1 GB of FASTQ (20 000 reads x 25 kb, G = 8e6)
set -e
python - <<PY
import sys, time; sys.path.insert(0,'.')
from goldrush_amd import synth
t=time.time(); synth.make_fastq('/tmp/big.fq', 8_000_000, 20000); print("fastq written in %.1fs" % (time.time()-t))
PY
ls -la /tmp/big.fq
mkdir -p /tmp/bigp
time goldrush_amd/bin/goldrush-path -k22 -w16 -t1000 -u5 -a1 -o0.1 -h3 -j16 -P0 -d5 -x10 -s1011011110110111101101 -g8e6 -b10 -r0.9 --silver_path -M5 -m20000 -i /tmp/big.fq -p /tmp/bigp/sp --verbose 2> /tmp/bigp/err.txt
grep -E "^in |inserting bit|assigning|Visited|Calculating min" /tmp/bigp/err.txt
