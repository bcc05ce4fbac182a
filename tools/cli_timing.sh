# developer tool: end-to-end timing of the goldrush-path CLI on a C0-like set
set -e
python - <<PY
import sys; sys.path.insert(0,'.')
from goldrush_amd import synth
synth.make_fastq('/tmp/c0.fq', 1_000_000, 2400)
PY
mkdir -p /tmp/c0p
time goldrush_amd/bin/goldrush-path -k22 -w16 -t1000 -u5 -a1 -o0.1 -h3 -j16 -P0 -d5 -x10 -s1011011110110111101101 -g1e6 -b10 -r0.9 --silver_path -M5 -m20000 -i /tmp/c0.fq -p /tmp/c0p/sp --verbose 2> /tmp/c0p/err.txt
grep -E "^in |inserting|assigning|Visited" /tmp/c0p/err.txt
cat /tmp/c0p/sp_*.fq > /tmp/c0p/all.fq
time goldrush_amd/bin/goldrush-path -k22 -w16 -t1000 -u5 -a1 -o0.1 -h3 -j16 -P0 -d5 -x10 -s1011011110110111101101 -g1e6 -b10 -m0 -i /tmp/c0p/all.fq -p /tmp/c0p/gp --verbose 2> /tmp/c0p/err2.txt
grep -E "^in |Visited" /tmp/c0p/err2.txt; ls -la /tmp/c0p/*.fa /tmp/c0p/*.fq | head
