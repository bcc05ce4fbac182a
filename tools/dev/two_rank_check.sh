#!/bin/bash
# GPU box: two ranks on the one GPU, C1 geometry, the counters both ranks agree on, several
# times per setting: "VAR=value,VAR=value" lists as arguments (default: the product's defaults)
port=29520
sets=("$@")
[ ${#sets[@]} -eq 0 ] && sets=("X=0")
for s in "${sets[@]}"; do
  for rep in 1 2 3; do
    port=$((port + 1))
    env $(echo $s | tr ',' ' ') python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --config C1 --steps 6 --no-cpu-baseline --backend gloo --share-gpu --reads 300000 --verify-ranks 2>&1 | grep -E "verify-ranks" | sed "s/^/$s /" | sed "s/verify-ranks: 2 ranks agree on//" | cut -c1-330
  done
done
