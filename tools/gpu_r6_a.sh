#!/bin/bash
# round 6, visit a: form 5 of the Stats kernel (fq_stats5.h) against form 4, the lane kernel's row prefetch into L2
# (FASTP_GPU_LANE_PREFETCH), per-kernel averages from rocprofv3 --kernel-trace --stats on 4 batches of 4,194,304 pairs
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_a.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6a}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "joint_table or every_quality or cells_at_their_capacity or plans_agree or at_baseline_scale or work_list or read_lengths or (equals_oracle and not scale)" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
trace() {   # NAME ENV... : kernel averages of one configuration
  NAME=$1; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/${V}_$NAME.log 2>&1
  python - "$V" "$NAME" "$@" >> $OUT <<'PY'
import csv, glob, sys, json
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(sys.argv[3:])}]"
if f:
    rows = {r["Name"]: r for r in csv.DictReader(open(f[0]))}
    for key in ("fq_lane_kernel", "fq_stats_kernel", "fq_stats5_kernel", "fq_reduce_kernel"):
        for n, r in rows.items():
            if key in n:
                line += f"  {key} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
try:
    j = json.loads(open(f"gpurun_out/{v}_{name}.log").read().strip().splitlines()[-1])
    line += f"  | step {j['ms_per_step']} ms, {j['value']} Mreads/s (under the tracer)"
except Exception as e:
    line += f"  | no bench line ({e})"
print(line)
PY
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
  tail -1 $OUT
}
trace form4              FASTP_GPU_STATS_V=4
trace form5              FASTP_GPU_VERBOSE=1
trace form5_nocyc        FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so FASTP_GPU_DEBUG_SKIP=64
trace form5_nokmer       FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so FASTP_GPU_DEBUG_SKIP=128
trace form5_noadds       FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so FASTP_GPU_DEBUG_SKIP=192
trace pf1                FASTP_GPU_LANE_PREFETCH=1
trace pf2                FASTP_GPU_LANE_PREFETCH=2
trace pf3                FASTP_GPU_LANE_PREFETCH=3
trace pf3_loads_only     FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so FASTP_GPU_LANE_PREFETCH=3 FASTP_GPU_DEBUG_SKIP=15
trace loads_only         FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so FASTP_GPU_DEBUG_SKIP=15
trace form5_again        FASTP_GPU_VERBOSE=1
grep -h "stats kernel" gpurun_out/${V}_form5.log | head -2 >> $OUT
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/${V}_bench_driver_cmd.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log | cut -c1-900
cat $OUT
