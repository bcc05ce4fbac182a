#!/bin/bash
# Run on the GPU box (gpurun): refresh the judged evidence for the current build.
#   tools/profile_round.sh <tag>      e.g. r01_v8
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench_default_flags.json 2> $out/${tag}_bench_default_flags.err
python bench.py --steps 112 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_full_stream.json 2> /dev/null
# kernel trace of the same default command (no CPU baseline leg: it only adds host time)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 bench.py --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
find $out/${tag}_prof -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
rm -rf $out/${tag}_prof
# two ranks on the one GPU (gloo): plumbing of the N > 1 path
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 12 --warmup 4 --no-cpu-baseline --backend gloo --share-gpu --reads 400000 > $out/${tag}_bench_2ranks_one_gpu_gloo_plumbing.json 2> $out/${tag}_2ranks.err
tail -2 $out/${tag}_2ranks.err
cat $out/${tag}_bench_default_flags.json $out/${tag}_bench_full_stream.json $out/${tag}_bench_under_rocprof.json $out/${tag}_bench_2ranks_one_gpu_gloo_plumbing.json
head -8 $out/${tag}_kernel_stats.csv
