#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box (ranks share cuda:0, gloo collectives
# with host staging; numbers are meaningless, the sharded-frame self-check is the point).
#   tools/rehearse_multirank.sh [nproc] [extra bench args...]
N=${1:-2}; shift || true
export VOLREND_BENCH_SHARE_GPU=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus $N --steps 32 --warmup 16 --batch 8 --no-cpu-baseline "$@"
