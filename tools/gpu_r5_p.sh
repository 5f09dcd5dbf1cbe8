#!/bin/bash
# round 5, visit p: the lane kernel's EXT >= 2 instantiations (fronts, -c, --merge) compiled for three (168 VGPRs, ~150 dwords spilled)
# or two (256 VGPRs, almost none) wavefronts per SIMD: -c and --merge lines with either
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 3 2; do echo "FASTP_GPU_LANE_EXT_WAVES=$v"; FASTP_GPU_LANE_EXT_WAVES=$v timeout 100 python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0), only='2x150 -f|2x150 -c|2x150 --merge'): print(json.dumps(r))
" 2>&1 | grep '^{' | cut -c1-230; done > gpurun_out/r5p_ext_waves.log 2>&1
cat gpurun_out/r5p_ext_waves.log
