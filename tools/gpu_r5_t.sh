#!/bin/bash
# round 5, visit t (the last GPU seconds): every flag set of the oracle comparison, merge on the lane plan, the three plans on the final Stats kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 38 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_gpu_equals_oracle or merge_on_the_lane or plans_agree" > gpurun_out/r5t_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5t_pytest.log
