#!/bin/bash
# multi-rank bench logic on a 1-GPU box (gloo test hook), nproc=1 nccl path, IVF timing
set -x
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 2048 --backend gloo 2>&1 | tail -3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 2 --warmup 1 --batch 4096 --no-cpu-baseline 2>&1 | tail -2
timeout 900 python scripts/bench_extra.py IVF_L IVF_S C2 --beams 8 --steps 2 2>&1 | tail -8
