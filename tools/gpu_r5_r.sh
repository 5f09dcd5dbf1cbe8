#!/bin/bash
# round 5, visit r: kernel trace of the -f 5 -F 5 line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o m -- python -c "
import sys, json, torch
sys.path.insert(0, '$GRAFT_REPO_ROOT'); sys.path.insert(0, '$GRAFT_REPO_ROOT/tools')
import os; os.chdir('$GRAFT_REPO_ROOT')
import bench
for r in bench.other_configs(torch.device('cuda', 0), only='2x150 -f'): print(json.dumps(r))
" > $GRAFT_REPO_ROOT/gpurun_out/r5r_run.log 2>&1; echo "rc=$?"
f=$(find /tmp/prof_r -name '*kernel_stats.csv' | head -1); echo "stats file: $f"
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r5r_front_kernel_stats.csv 2>/dev/null
head -14 "$f" | cut -d, -f1-6
