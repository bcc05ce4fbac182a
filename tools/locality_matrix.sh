#!/bin/bash
# Round 4: the memory system's random-access rates by footprint, access shape and locality
# (tools/gather_bench.hip modes 10-19) -> gpurun_out/<tag>_locality_matrix.txt
# usage (GPU box): tools/locality_matrix.sh <tag>
tag=${1:-r04}
out=gpurun_out/${tag}_locality_matrix.txt
mkdir -p gpurun_out
cd "$(dirname "$0")" && B=./gather_bench
[ -x $B ] && [ $B -nt gather_bench.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
cd ..
B=tools/gather_bench
{
echo "# random 64-B lines read by a quad (mode 10, what k_query does) by table size"
for mib in 8 32 64 128 256 512 1024 4096 16384 65536; do timeout 120 $B $mib 64 64 10 2; done
echo "# random 4-B loads, one per lane (mode 1) by table size"
for mib in 4 16 32 64 128 256 1024 16384; do timeout 120 $B 64 $mib 64 1 4; done
echo "# Bloom-style bit look-ups (mode 19) by table size"
for mib in 4 32 128 512; do timeout 120 $B 64 $mib 64 19 4; done
echo "# random atomicOr u32 (mode 12) by table size"
for mib in 4 32 128 512 16384; do timeout 120 $B 64 $mib 64 12 4; done
echo "# random atomicCAS u64 (mode 13) by table size"
for mib in 64 256 1024 16384; do timeout 120 $B 64 $mib 64 13 4; done
echo "# random 1-byte stores (mode 14)"
for mib in 128 16384; do timeout 120 $B 64 $mib 64 14 4; done
echo "# 128-B buckets read by 8 lanes (mode 15) / 64-B line read + 4 B written back (mode 16), 16 GiB"
timeout 120 $B 16384 64 64 15 2
timeout 120 $B 16384 64 64 16 2
echo "# reference for the binned modes: random lines in their loop shape (mode 20), 16 GiB and 64 GiB"
timeout 120 $B 16384 64 64 20 8
timeout 120 $B 65536 64 64 20 8
timeout 120 $B 65536 64 64 20 16
echo "# page-binned sweep in address order (mode 11), 16 GiB table, density 0.6 lines drawn per line"
for r in 64 1024 2048 8192 65536 2097152 67108864 1073741824 17179869184; do GB_REGION=$r timeout 120 $B 16384 64 64 11 8; done
echo "# the same at density 0.15 (a quarter of a window's probes per sweep)"
for r in 2048 8192 2097152 67108864; do GB_REGION=$r GB_DENSITY=0.15 timeout 120 $B 16384 64 64 11 8; done
echo "# the same, 16 loads in flight per lane"
for r in 64 2048 2097152 17179869184; do GB_REGION=$r timeout 120 $B 16384 64 64 11 16; done
echo "# the 16 lines of a wave step (mode 17) / the 64 lines of a workgroup step (mode 18) inside one random region, 64 GiB table"
for m in 17 18; do for r in 1024 2048 8192 65536 2097152; do GB_REGION=$r timeout 120 $B 65536 64 64 $m 8; done; done
echo "# reference: mode 10 on 64 GiB, unroll 2 / 4"
timeout 120 $B 65536 64 64 10 2
timeout 120 $B 65536 64 64 10 4
} > $out 2>&1
cat $out
