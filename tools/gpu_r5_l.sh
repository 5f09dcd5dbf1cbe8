#!/bin/bash
# round 5, visit l: --merge on the lane plan - its GPU cases (oracle, reference goldens, stress, the forced slow path), the line of
# the other configurations
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 270 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "merge_on_the_lane or merge_stress or (equals_oracle and merge) or (reference_golden and merge)" > gpurun_out/r5l_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5l_pytest.log
timeout 120 python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0), only='--merge'): print(json.dumps(r))
" > gpurun_out/r5l_merge_line.log 2>&1; echo "merge line rc=$?"
grep '^{' gpurun_out/r5l_merge_line.log | cut -c1-300
