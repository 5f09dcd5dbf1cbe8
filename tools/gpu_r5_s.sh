#!/bin/bash
# round 5, visit s (three short runs): the Stats kernel with a front trim - the mode character from a kept base (no change), the
# front 5-mers by grouped loads (-f 5 -F 5: 4.51 -> 3.40 ms) - and the correction lists' two atomics in one round trip (no change)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 60 python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/r5s_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r5s_bench.log | cut -c1-1200
timeout 60 python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0), only='2x150 -f|2x150 --umi|2x150 -c|2x150 --merge'): print(json.dumps(r))
" 2>&1 | grep '^{' | cut -c1-230 > gpurun_out/r5s_lines.log; cat gpurun_out/r5s_lines.log
