#!/bin/bash
# short GPU-box visit while iterating on the fused kernel: a parity subset, a short bench, SQ counters, phase shares
#   tools/gpu_quick.sh TAG ["pytest -k expression"]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-q}
KEXPR=${2:-"test_gpu_equals_oracle or bit_positions or one_gap or stress or read_lengths or launches"}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$KEXPR" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log
timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_$TAG.log
bash tools/pmc.sh $TAG > gpurun_out/pmc_${TAG}.txt 2>&1; tail -24 gpurun_out/pmc_${TAG}.txt
timeout 120 python tools/phase_prof.py > gpurun_out/phase_$TAG.log 2>&1; tail -1 gpurun_out/phase_$TAG.log
