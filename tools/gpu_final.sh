#!/bin/bash
# last visit of the round: the whole -m gpu suite, smoke, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest.log | tail -2
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2>&1) 2>&1 | grep real; echo "bench rc=$?"
tail -1 gpurun_out/bench_driver.log | cut -c1-300
