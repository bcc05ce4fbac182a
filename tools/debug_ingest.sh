set -e
python - <<PY
import sys; sys.path[:0]=['.','tests','oracle']
import test_gpu_cli as t
t._mk_fastq('/tmp/dbg.fq', 300_000, 900, 6000, 4000, seed=5, lower=True, with_n=50, short=40)
PY
mkdir -p /tmp/dbg_h /tmp/dbg_g
A="-k22 -w16 -t500 -u5 -a1 -o0.1 -h3 -j8 -d5 -x10 -s1011011110110111101101 -g300000 -b4 -H4000000 -P0 -r0.9 --silver_path -M3 -m3500 -i /tmp/dbg.fq --verbose"
GRP_HOST_INGEST=1 goldrush_amd/bin/goldrush-path $A -p /tmp/dbg_h/o 2> /tmp/dbg_h/err
goldrush_amd/bin/goldrush-path $A -p /tmp/dbg_g/o 2> /tmp/dbg_g/err
diff <(grep -v "^in " /tmp/dbg_h/err) <(grep -v "^in " /tmp/dbg_g/err) | head -20
for f in o_1.fq o_2.fq o_3.fq; do cmp /tmp/dbg_h/$f /tmp/dbg_g/$f | head -2; done
diff <(grep "^@" /tmp/dbg_h/o_1.fq | head -50) <(grep "^@" /tmp/dbg_g/o_1.fq | head -50) | head
