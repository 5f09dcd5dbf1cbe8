#!/bin/bash
# pending parity tests + the lane kernel with and without its scheduling fences (libfastp_gpu_nf.so = -DFQ_LANE_NO_FENCE)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-lane3}
OUT=gpurun_out/lane_ab_$TAG.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_binding.py -m gpu -x -q -k "at_scale or auto_adapter" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log
run() { NAME=$1; shift; env "$@" timeout 300 python bench.py --steps 32 --warmup 8 --batches 8 --no-cpu --no-extras > gpurun_out/ab_${TAG}_$NAME.log 2>&1; tail -1 gpurun_out/ab_${TAG}_$NAME.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$NAME', j['value'], 'Mreads/s kernels', r['kernel_avg_ms'], 'ms per', r['pairs_per_launch'])" | tee -a $OUT; }
for rep in 1 2; do
run fence_default
run fence_nostats FASTP_GPU_DEBUG_SKIP=16
if [ -f fastp_amd/libfastp_gpu_nf.so ]; then
  run nf_default FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_nf.so
  run nf_nostats FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_nf.so FASTP_GPU_DEBUG_SKIP=16
fi
done
