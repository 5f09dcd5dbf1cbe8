#!/bin/bash
# round 4, first GPU visit: the stream binding on the real library (binding tests + the drop-in bench)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ref_binding.py -m gpu -x -q > gpurun_out/pytest_binding.log 2>&1; echo "binding pytest rc=$?"; tail -3 gpurun_out/pytest_binding.log
timeout 1500 python tools/dropin_bench.py --pairs 4000000 --big 16000000 > gpurun_out/r04_dropin.txt 2>&1; echo "dropin rc=$?"
cat gpurun_out/r04_dropin.txt
