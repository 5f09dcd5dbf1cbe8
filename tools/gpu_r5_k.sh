#!/bin/bash
# round 5, visit k: the text kernel BESIDE the lane kernel (FASTP_GPU_EXACT_EARLY) and mOverRepSeqDist as a difference array
# (FASTP_GPU_OVR_DIFF): the GPU cases that cover them, each switch off / on for the configuration it is about, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "exotic or text_kernel or overrep or several_launches or option_fuzz" > gpurun_out/r5k_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5k_pytest.log
oc() {
timeout 200 python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0), only='$1'): print(json.dumps(r))
" 2>&1 | grep '^{' | cut -c1-330
}
for v in 0 1; do echo "FASTP_GPU_EXACT_EARLY=$v"; FASTP_GPU_EXACT_EARLY=$v oc "soft-masked"; done > gpurun_out/r5k_exact_early.log 2>&1
cat gpurun_out/r5k_exact_early.log
for v in 0 1; do echo "FASTP_GPU_OVR_DIFF=$v"; FASTP_GPU_OVR_DIFF=$v oc "configs[4]"; done > gpurun_out/r5k_ovr_diff.log 2>&1
cat gpurun_out/r5k_ovr_diff.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5k_bench_driver_cmd.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r5k_bench_driver_cmd.log | cut -c1-700
