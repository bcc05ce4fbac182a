#!/bin/bash
# Round 5: what one record of k_batch_collect costs the memory system by where the rank's count / claim lives
# (tools/gather_bench.hip modes 21-26) -> gpurun_out/<tag>_collect_matrix.txt
# usage (GPU box): tools/collect_matrix.sh <tag>
tag=${1:-r05}
out=gpurun_out/${tag}_collect_matrix.txt
mkdir -p gpurun_out
cd "$(dirname "$0")" && B=./gather_bench
[ -x $B ] && [ $B -nt gather_bench.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
cd ..
B=tools/gather_bench
{
echo "# today (mode 21): header of a 64-B bucket (A) -> CAS u64 on a random count word (B) -> ID written back; C2-sized tables, unroll 1/2/4"
for u in 1 2 4; do timeout 300 $B 65536 49152 64 21 $u; done
echo "# count line in a second bucket-indexed table (mode 26: same index, another allocation)"
for u in 1 2 4; do timeout 300 $B 65536 65536 64 26 $u; done
echo "# 128-B unit (mode 22): header + claim word loaded together, CAS on the claim, atomicAdd on the count, ID written back"
for u in 1 2 4; do timeout 300 $B 131072 64 64 22 $u; done
echo "# 128-B unit, claim and count in one word (mode 23)"
for u in 1 2 4; do timeout 300 $B 131072 64 64 23 $u; done
echo "# 128-B unit read whole by 8 lanes, 4 B written into each half (mode 25)"
for u in 1 2; do timeout 300 $B 131072 64 64 25 $u; done
echo "# the query's quad read of a 64-B line: lines 64 B apart in 64 GiB (mode 10) / 128 B apart in 128 GiB (mode 24)"
for u in 2 4; do timeout 300 $B 65536 64 64 10 $u; timeout 300 $B 131072 64 64 24 $u; done
echo "# the same at C1 size (16 GiB / 32 GiB)"
timeout 300 $B 16384 64 64 10 2; timeout 300 $B 32768 64 64 24 2
timeout 300 $B 16384 8192 64 21 2; timeout 300 $B 32768 64 64 22 2; timeout 300 $B 32768 64 64 23 2
echo "# C4 size: 109 GB of buckets today, 218 GB as units"
timeout 300 $B 106496 64 64 10 2; timeout 300 $B 212992 64 64 24 2
echo "# the count line GB_L1OFF bytes behind its bucket line (mode 30; 64 = the unit): is it the DRAM page or the translation that is shared?"
for off in 64 256 4096 65536 2097152 1073741824; do echo -n "off $off: "; GB_L1OFF=$off timeout 300 $B 131072 64 64 30 2; done
echo "# ... and the query's quad read over those layouts (mode 31)"
for off in 64 4096 2097152 1073741824; do echo -n "off $off: "; GB_L1OFF=$off timeout 300 $B 131072 64 64 31 2; done
echo "# one record of the collect pass by its parts (mode 50, GB_SHAPE bits: 1 no load in front of the CAS, 2 ID by atomicExch, 4 ID read in front of the CAS, 8 ID stored write-through, 16 no record words, 32 header by agent-scope loads, 64 no ID store)"
for sh in 0 1 2 3 4 5 8 16 32 64 65; do echo -n "shape $sh: "; GB_SHAPE=$sh timeout 300 $B 131072 4096 64 50 1; done
echo "# one record per lane and workgroup (524288 workgroups) against 64 per lane (8192): the launch shape is not the limit"
timeout 300 $B 131072 64 1 23 1 524288; timeout 300 $B 131072 64 64 23 1 8192
} > $out 2>&1
cat $out
