#!/bin/bash
# N ranks on ONE GPU over gloo: rehearses the WHOLE `bench.py --gpus N` control flow the driver's SCALE run takes (not a performance
# number: the ranks share one card and gloo carries the collectives).  N <= 6: a GPU box allows at most 6 processes on its card, so the
# 8-rank flow itself cannot be rehearsed on one card; torchrun's agent holds the card open too, so N <= 4 in practice (5 ranks + agent + one more process were counted as 7): four ranks exercise the same code with
#   bash profiles/tools/rehearse_nrank.sh 4 > gpurun_out/bench_gloo_4ranks.json
N=${1:-4}
export IONO_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --steps 8 --warmup 2 --extras-timeout 900
