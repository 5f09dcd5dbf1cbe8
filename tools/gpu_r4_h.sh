#!/bin/bash
# round 4: the text kernel (units with letters outside ACGTN) on the GPU: parity tests, the binding, what it costs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "random_option or (exotic and (binding or stream or patched))" > gpurun_out/pytest_exotic2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_exotic2.log
timeout 300 python tools/exotic_bench.py > gpurun_out/r04_exotic_cost.txt 2>&1; echo "bench rc=$?"; cat gpurun_out/r04_exotic_cost.txt
