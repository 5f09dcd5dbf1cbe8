#!/bin/bash
# lane kernel staging: all vectors of a stage in flight (batch 10, default) / batch 5 (libfastp_gpu_nf.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-ab8}
OUT=gpurun_out/ab_$TAG.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "equals_oracle or plans_agree or read_lengths" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_$TAG.log
run() { NAME=$1; shift; env "$@" timeout 300 python bench.py --steps 48 --warmup 8 --batches 8 --no-cpu --no-extras > gpurun_out/ab_${TAG}_$NAME.log 2>&1; tail -1 gpurun_out/ab_${TAG}_$NAME.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$NAME', j['value'], 'Mreads/s step', j['ms_per_step'], 'kernels', r['kernel_avg_ms'], 'ms per', r['pairs_per_launch'])" | tee -a $OUT; }
for rep in 1 2; do
run batch10
run batch10_nostats FASTP_GPU_DEBUG_SKIP=16
run batch5 FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_nf.so
run batch5_nostats FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_nf.so FASTP_GPU_DEBUG_SKIP=16
done
run batch10_4wg FASTP_GPU_LANE_BLOCKS_PER_CU=4
run batch10_2wg FASTP_GPU_LANE_BLOCKS_PER_CU=2
