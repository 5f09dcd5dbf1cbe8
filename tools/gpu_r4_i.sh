#!/bin/bash
# round 4, final tree: the whole -m gpu suite, smoke, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" gpurun_out/pytest_final.log | tail -2
timeout 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2>&1; echo "bench (driver's command) rc=$?"; tail -1 gpurun_out/bench_driver_cmd.log | cut -c1-6000
