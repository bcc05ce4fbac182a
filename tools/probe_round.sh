#!/bin/bash
# GPU box: host facts, the random-line ceiling, one default bench run with the per-slice trace.
#   tools/probe_round.sh <tag>
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
{ nproc; free -g; lscpu | grep -E "Model name|Socket|Thread|Core"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; df -h /dev/shm /tmp | cat; rocm-smi --showmeminfo vram | cat; } > $out/${tag}_host_info.txt 2>&1
timeout 900 python3 tools/gather_ceiling.py $out/${tag}_gather_ceiling.json > $out/${tag}_gather_ceiling.log 2>&1
timeout 1500 python3 bench.py --trace "${@:2}" > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err
tail -3 $out/${tag}_bench_default_flags.err
cat $out/${tag}_host_info.txt
cat $out/${tag}_bench_default_flags.json
