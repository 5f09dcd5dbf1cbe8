#!/bin/bash
# round 6, visit u: fq_dup_losers_kernel on the launch stream in front of the Stats kernel (product) against visit t's order
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_u.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6u}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "dup or dedup or exotic or baseline_scale or plans_agree or shard or fuzz" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
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
line product            FASTP_GPU_VERBOSE=0
line visit_t_order      FASTP_GPU_DUP_LOSERS_FIRST=0
line losers_fold_first  FASTP_GPU_MISC_FOLD_FIRST=1
line product_2          FASTP_GPU_VERBOSE=0
line visit_t_order_2    FASTP_GPU_DUP_LOSERS_FIRST=0
line losers_fold_first_2 FASTP_GPU_MISC_FOLD_FIRST=1
for T in product:FASTP_GPU_VERBOSE=0 fold_first:FASTP_GPU_MISC_FOLD_FIRST=1; do
  N=${T%%:*}; E=${T#*:}
  rm -rf gpurun_out/prof/${V}_step_$N
  env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/${V}_step_$N -o t -- python bench.py --steps 24 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/${V}_step_$N.log 2>&1; echo "step trace $N rc=$?"
  (echo "== $N [$E]"; python tools/step_timeline.py gpurun_out/prof/${V}_step_$N 2) >> gpurun_out/${V}_step_timeline.txt 2>&1
  find gpurun_out/prof/${V}_step_$N -name "*_kernel_trace.csv" -delete
done
cat gpurun_out/${V}_step_timeline.txt
