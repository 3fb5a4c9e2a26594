#!/bin/bash
# 2 ranks on ONE GPU over gloo: rehearses bench.py's multi-rank control flow (not a performance number)
export IONO_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 2
