#!/bin/bash
# round 4: the final tree's -m gpu suite + smoke, the driver's bench command, the 100 M-pair stream with several reader settings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest.log | head -2
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2>&1; echo "bench (driver's command) rc=$?"; tail -1 gpurun_out/bench_driver_cmd.log | cut -c1-3000
timeout 400 python tools/e2e_dropin_100M.py --pairs 100000000 --no-ref > gpurun_out/r04_dropin_100M_reader_ab.txt 2>&1; echo "100M reader A/B rc=$?"
cat gpurun_out/r04_dropin_100M_reader_ab.txt | cut -c1-500
