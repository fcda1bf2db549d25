#!/bin/bash
# N ranks on ONE GPU through gloo: checks the multi-rank bench path end to end (not a performance number).
n=${1:-2}
RT_BENCH_REHEARSE=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 20 --warmup 3 2>&1
