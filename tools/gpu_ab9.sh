#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-ab9}
OUT=gpurun_out/ab_$TAG.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "equals_oracle or plans_agree or read_lengths or polyg or stress" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_$TAG.log
run() { NAME=$1; shift; env "$@" timeout 300 python bench.py --steps 48 --warmup 8 --batches 8 --no-cpu --no-extras > gpurun_out/ab_${TAG}_$NAME.log 2>&1; tail -1 gpurun_out/ab_${TAG}_$NAME.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$NAME', j['value'], 'Mreads/s step', j['ms_per_step'], 'kernels', r['kernel_avg_ms'], 'ms per', r['pairs_per_launch'])" | tee -a $OUT; }
run default
run nostats FASTP_GPU_DEBUG_SKIP=16
run default_b
run nostats_b FASTP_GPU_DEBUG_SKIP=16
timeout 600 python bench.py --steps 24 --warmup 4 --batches 8 --no-cpu > gpurun_out/ab_${TAG}_extras.log 2>&1
tail -1 gpurun_out/ab_${TAG}_extras.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps(j.get('other_configs')), json.dumps(j.get('e2e_gpu')), json.dumps(j.get('e2e_dropin')))" | tee -a $OUT
