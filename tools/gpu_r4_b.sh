#!/bin/bash
# round 4, second GPU visit: stream C ABI + binding on the real library, the drop-in bench, kernel timings of the new lane kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stream_abi.py tests/test_ref_binding.py -m gpu -q -x > gpurun_out/pytest_stream.log 2>&1; echo "stream+binding pytest rc=$?"; tail -3 gpurun_out/pytest_stream.log
timeout 200 python bench.py --steps 20 --warmup 2 --no-cpu --no-extras > gpurun_out/bench_steps20.log 2>&1; echo "bench --steps 20 rc=$?"; tail -1 gpurun_out/bench_steps20.log | cut -c1-1200
for SK in 1 4 8 13; do
  FASTP_GPU_DEBUG_SKIP=$SK timeout 120 python bench.py --steps 6 --warmup 2 --batches 2 --pairs 4194304 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('DEBUG_SKIP=$SK kernel_avg_ms', j['roofline']['kernel_avg_ms'], 'ms_per_step', j['ms_per_step'])"
done > gpurun_out/r04_lane_ablation.txt 2>&1
cat gpurun_out/r04_lane_ablation.txt
timeout 900 python tools/dropin_bench.py --pairs 4000000 --big 12000000 > gpurun_out/r04_dropin.txt 2>&1; echo "dropin rc=$?"
cat gpurun_out/r04_dropin.txt | cut -c1-420
