#!/bin/bash
# round 6, visit x: the lane kernel's chunk pool (the last part of a launch's chunks belongs to no workgroup; second run: asks prefetched, a 32nd by default) against no pool / other sizes
#   gpurun --timeout 1800 -- 'bash tools/gpu_r6_x.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6x}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "test_gpu_equals_oracle or baseline_scale or plans_agree or scale_config or fuzz or golden" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
line() {   # NAME ENV... : the bench's own line (720 steps of 4 Mi pairs) under the switches
  NAME=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu > gpurun_out/${V}_$NAME.log 2>&1
  python - "$NAME" "$*" gpurun_out/${V}_$NAME.log >> $OUT <<'PY'
import json, sys
name, sw, path = sys.argv[1:4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(f"{name:24s} [{sw}]  {d['value']:8.1f} Mreads/s  {d['ms_per_step']:.4f} ms per step of {d['config']['pairs_per_step_per_gpu']} pairs  kernels {d['roofline']['kernel_avg_ms']:.4f} ms  frac {d['roofline']['frac']:.5f}")
except Exception as e:
    print(f"{name:24s} [{sw}]  failed: {e!r}")
PY
  tail -1 $OUT
}
cfg() {   # NAME CONFIG ENV... : one line of other_configs
  NAME=$1; CFG=$2; shift; shift
  env "$@" timeout 300 python tools/one_config.py "$CFG" > gpurun_out/${V}_$NAME.log 2>&1
  echo "$NAME [$*] $(grep '^{' gpurun_out/${V}_$NAME.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['frac'])")" >> $OUT
  tail -1 $OUT
}
line pool32      FASTP_GPU_VERBOSE=0
line nopool      FASTP_GPU_LANE_POOL_LOG2=0
line pool16      FASTP_GPU_LANE_POOL_LOG2=4
line pool64      FASTP_GPU_LANE_POOL_LOG2=6
line pool128     FASTP_GPU_LANE_POOL_LOG2=7
line pool32_2    FASTP_GPU_VERBOSE=0
line nopool_2    FASTP_GPU_LANE_POOL_LOG2=0
line pool64_2    FASTP_GPU_LANE_POOL_LOG2=6
cfg se_pool32    "configs[1]" FASTP_GPU_VERBOSE=0
cfg se_nopool    "configs[1]" FASTP_GPU_LANE_POOL_LOG2=0
cfg se_pool64    "configs[1]" FASTP_GPU_LANE_POOL_LOG2=6
cfg c_pool32     " -c "       FASTP_GPU_VERBOSE=0
cfg c_nopool     " -c "       FASTP_GPU_LANE_POOL_LOG2=0
cfg c4_pool32    "configs[4]" FASTP_GPU_VERBOSE=0
cfg c4_nopool    "configs[4]" FASTP_GPU_LANE_POOL_LOG2=0
cfg se_pool32_2  "configs[1]" FASTP_GPU_VERBOSE=0
cfg se_nopool_2  "configs[1]" FASTP_GPU_LANE_POOL_LOG2=0
cfg se_pool64_2  "configs[1]" FASTP_GPU_LANE_POOL_LOG2=6
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/${V}_bench_driver_cmd.log 2>&1; echo "bench (driver's command, no extras) rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log | cut -c1-300
grep -o '"roofline": {[^}]*}' gpurun_out/${V}_bench_driver_cmd.log | tail -1
