#!/bin/bash
# the N > 1 code path of bench.py rehearsed on ONE GPU: two ranks share device 0, gloo carries the collectives
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --batches 3 --pairs 1048576 > gpurun_out/n2.log 2>&1; echo "n2 rc=$?"
tail -3 gpurun_out/n2.log | cut -c1-1200
