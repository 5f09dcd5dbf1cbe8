#!/bin/bash
# round 5, last visit: the one case the full suite failed (fixed), the cases with letters outside ACGTN with the text kernel
# beside the Stats kernel, the other configurations, the driver's bench command on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider -k "tile_shapes or exotic or text_kernel or several_launches or streamed_batches" > gpurun_out/r5j_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5j_pytest.log
timeout 300 python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0)): print(json.dumps(r))
" > gpurun_out/r5j_other_configs.log 2>&1; echo "other configs rc=$?"
grep '^{' gpurun_out/r5j_other_configs.log | cut -c1-250
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5j_bench_driver_cmd.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r5j_bench_driver_cmd.log | cut -c1-900
