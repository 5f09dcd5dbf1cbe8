#!/bin/bash
# A/B of the kernel plans on one GPU box: fused (one 1024-lane workgroup per CU, Stats inside) vs split (256-lane
# per-read workgroups + streaming Stats kernel), a sweep of the split plan's geometry, a rocprofv3 kernel trace of
# the default split configuration.   tools/gpu_split_ab.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-ab}
OUT=gpurun_out/split_ab_$TAG.txt
: > $OUT
run() {  # name, env...
  NAME=$1; shift
  env "$@" timeout 300 python bench.py --steps 48 --warmup 8 --batches 8 --no-cpu > gpurun_out/ab_${TAG}_$NAME.log 2>&1
  python - "$NAME" gpurun_out/ab_${TAG}_$NAME.log >> $OUT <<'PY'
import sys, json
name, path = sys.argv[1:3]
try:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{name:34s} {j['value']:9.1f} Mreads/s  step {j['ms_per_step']:.3f} ms  kernels {r['kernel_avg_ms']:.4f} ms per {r['pairs_per_launch']} pairs  frac {r['frac']}")
except Exception as e:
    print(f"{name:34s} FAILED {e!r}: " + open(path).read()[-300:].replace("\n", " | "))
PY
  tail -1 $OUT
}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "equals_oracle or one_gap or stress or read_lengths or launches or golden" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log
run fused_1024 FASTP_GPU_SPLIT=0
run split_default FASTP_GPU_VERBOSE=1
run split_lds32 FASTP_GPU_LDS_KB=32
run split_lds48 FASTP_GPU_LDS_KB=50
run split_t128 FASTP_GPU_THREADS=128 FASTP_GPU_LDS_KB=20
run split_st256 FASTP_GPU_STATS_THREADS=256
run split_st1024 FASTP_GPU_STATS_THREADS=1024
run split_nostats FASTP_GPU_DEBUG_SKIP=16
run split_statsonly_noatom FASTP_GPU_DEBUG_SKIP=448
grep -h "fastp_gpu:" gpurun_out/ab_${TAG}_split_default.log | head -2 >> $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${TAG}_split -o trace -- python bench.py --steps 24 --warmup 2 --batches 4 --no-cpu > gpurun_out/rocprof_${TAG}.log 2>&1; echo "trace rc=$?"
python - gpurun_out/prof/${TAG}_split >> $OUT <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    print("== " + f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12: print(",".join(row[:8]))
PY
cat $OUT
