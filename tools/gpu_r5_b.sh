#!/bin/bash
# round 5, first GPU visit: the WHOLE -m gpu suite (no -x: every failure is seen in one visit; --durations for the sizing),
# smoke, the driver's bench command.   gpurun --timeout 1700 -- 'bash tools/gpu_r5_b.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1350 python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider > gpurun_out/r5b_pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
tail -60 gpurun_out/r5b_pytest.log | cut -c1-300
timeout 200 python __graft_entry__.py smoke > gpurun_out/r5b_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5b_smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5b_bench_driver_cmd.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r5b_bench_driver_cmd.log | cut -c1-6000
