#!/bin/bash
# round 5, visit m: the overrepresentation analysis on the tail stream beside the Stats kernel (FASTP_GPU_OVR_TAIL): its GPU cases,
# the switch off / on for configs[4]'s per-GPU share
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "overrep or config5" > gpurun_out/r5m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5m_pytest.log
for v in 0 1; do echo "FASTP_GPU_OVR_TAIL=$v"; FASTP_GPU_OVR_TAIL=$v timeout 100 python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0), only='configs[4]'): print(json.dumps(r))
" 2>&1 | grep '^{' | cut -c1-300; done > gpurun_out/r5m_ovr_tail.log 2>&1
cat gpurun_out/r5m_ovr_tail.log
