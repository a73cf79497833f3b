#!/bin/bash
# single-rank plumbing check of the expert-parallel bench path (the multi-GPU runs are the driver's)
set -u
for mode in a2a ar; do
  echo "== --force-ep --ep-mode $mode"
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 150 python bench.py --gpus 1 --steps 50 --warmup 5 --force-ep --ep-mode $mode --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-330
done
echo "== torchrun 1 proc"; timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-200
