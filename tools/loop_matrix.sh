#!/bin/bash
# GPU box: the head of the C1 / C2 streams under the loop's developer switches.
export TMPDIR=/tmp
run() { # name, config, reads, env...
  local name=$1 cfg=$2 reads=$3; shift 3
  echo "== $name $cfg $*"
  env "$@" timeout 600 python3 bench.py --config $cfg --no-cpu-baseline --reads $reads --steps 4 --trace 2>&1 >/dev/null | grep "^reads" | head -3
}
for cfg in "C1 200000" "C2 100000"; do
  set -- $cfg
  run windows $1 $2 GRP_LOOP=off
  run default $1 $2 GRP_X=1
  run whole_d1 $1 $2 GRP_LOOP_UNIT=1 GRP_LOOP_MIN_DEPTH=1
  run whole_d4 $1 $2 GRP_LOOP_UNIT=1 GRP_LOOP_MIN_DEPTH=4
  run whole_d8 $1 $2 GRP_LOOP_UNIT=1 GRP_LOOP_MIN_DEPTH=8
  run units128 $1 $2 GRP_LOOP_UNIT=128
  run units256_d2 $1 $2 GRP_LOOP_UNIT=256 GRP_LOOP_MIN_DEPTH=2
done
