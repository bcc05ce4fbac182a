#!/bin/bash
# GPU box: kernel stats of the previous tree (.ab_old) and of this one, same box     tools/dev/r6_prof_old_new.sh <tag> "<env>"
tag=${1:-r06_p}
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for d in .ab_old .; do
  n=new; [ "$d" = ".ab_old" ] && n=old
  (cd $d && env $2 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${n}_prof -o ${tag} -- python3 bench.py --no-cpu-baseline --no-pipeline-shaped > $out/${tag}_${n}.json 2> $out/${tag}_${n}.err)
  find $out/${tag}_${n}_prof -name "*kernel_stats.csv" -exec cp {} $out/${tag}_${n}_kernel_stats.csv \;
  rm -rf $out/${tag}_${n}_prof
  echo "== $n"; head -8 $out/${tag}_${n}_kernel_stats.csv | cut -c1-260
done
