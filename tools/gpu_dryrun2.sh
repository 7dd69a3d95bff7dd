#!/bin/bash
# N > 1 protocol dry run on ONE GPU: 2 ranks share the device, key lists staged through gloo
cd $GRAFT_REPO_ROOT
SE_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 40 --warmup 6 --no-cpu-baseline 2>&1 | tail -3
