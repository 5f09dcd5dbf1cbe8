#!/bin/bash
# round 5: -c on the lane plan on the hardware: the correction goldens / oracle parity / fuzz, then the other configurations
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "correction or corr or random_option or plans_agree or umi or trim_fixed or dedup" > gpurun_out/r5g_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5g_pytest.log
timeout 300 python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0)): print(json.dumps(r))
" > gpurun_out/r5g_other_configs.log 2>&1; echo "other configs rc=$?"
grep '^{' gpurun_out/r5g_other_configs.log
