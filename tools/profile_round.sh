#!/bin/bash
# Run on the GPU box (gpurun): refresh the judged evidence for the current build.
#   tools/profile_round.sh <tag>      e.g. r02_v13
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
# the random-line ceiling of this box's memory system (bench.py reads the newest one in profiles/)
timeout 600 python3 tools/gather_ceiling.py $out/${tag}_gather_ceiling.json > /dev/null 2>&1
cp $out/${tag}_gather_ceiling.json profiles/ 2>/dev/null
# HBM bytes per probe of THIS build (bench.py only applies a summary whose csrc hash matches): first, so that
# the default line below carries it
tools/pmc_round.sh ${tag} > $out/${tag}_pmc.log 2>&1
cp $out/${tag}_pmc_summary.json profiles/ 2>/dev/null
# the driver's command: C2, whole stream, with the like-for-like CPU baseline
python3 bench.py > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err
# C1 (BASELINE configs[1]), whole stream
python3 bench.py --config C1 --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_bench_c1_full_stream.json 2> /dev/null
tools/rocprof_round.sh ${tag}
# the insert-heavy head alone (first 300 k reads of C2: batches): slice trace + per-kernel stats
tools/head_profile.sh ${tag} 300000 > $out/${tag}_head_trace.txt 2>&1
# C4 geometry (h = 5, silver mode, 5 paths): the stream is head after head
python3 bench.py --config C4 --reads 4000000 --no-cpu-baseline > $out/${tag}_bench_c4_4M.json 2> /dev/null
# exactness at full size: C2's head (61 Gbit filter, 700 k reads) through the default path twice and
# through the classic windows only; one rank against two ranks (C1 and C2): the run's counters
tools/dev/mode_truth_check.sh C2 1500000 > $out/${tag}_truth_modes_c2.txt 2>&1
(tools/dev/rank_truth_check.sh C1 300000 ${tag}; tools/dev/rank_truth_check.sh C2 600000 ${tag}) > $out/${tag}_truth_ranks.txt 2>&1
# the N = 8 time model from this build's default line and the measured 1-rank / 2-rank pair (counted constants)
python3 tools/scale_model.py $out/${tag}_bench_default_flags.json --ranks $out/${tag}_ranks_C1_one.json $out/${tag}_ranks_C1_two.json > $out/${tag}_scale_model.txt 2>&1
# a repeat-rich genome (40 % repeat families): golden pass + pipeline-shaped pass, aux.repeats
python3 bench.py --repeat-frac 0.4 --no-cpu-baseline > $out/${tag}_bench_repeats.json 2> /dev/null
# what a record of the collect pass costs by where its count lives (gather_bench modes 21-37, 50)
tools/collect_matrix.sh ${tag} > /dev/null 2>&1
# the drop-in binary end to end: 100 k reads of C1 as a 5 GB FASTQ, silver and golden, reads kept on the device / second parse
timeout 1500 python3 tools/cli_end_to_end.py $out/${tag}_cli_end_to_end.json 100000 > $out/${tag}_cli_e2e.log 2>&1
# two ranks on the one GPU (gloo): plumbing of the N > 1 path (fill merge, striped windows, shm exchange)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --config C1 --steps 6 --no-cpu-baseline --backend gloo --share-gpu --reads 300000 --verify-ranks > $out/${tag}_bench_2ranks_one_gpu_gloo_plumbing.json 2> $out/${tag}_2ranks.err
grep verify-ranks $out/${tag}_2ranks.err | head -1
cat $out/${tag}_bench_default_flags.json $out/${tag}_bench_c1_full_stream.json $out/${tag}_bench_under_rocprof.json
head -12 $out/${tag}_kernel_stats.csv
