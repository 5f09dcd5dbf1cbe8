#!/bin/bash
# round 5, visit n: the stream binding's and the patched reference's merge cases on the GPU (merge mode is on the lane plan now)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 170 python -m pytest tests/test_stream_abi.py tests/test_ref_binding.py -m gpu -q -p no:cacheprovider -x -k "merge and (stream_equals_reference_golden or patched_reference_equals_reference)" > gpurun_out/r5n_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5n_pytest.log
