#!/bin/bash
# one GPU-box visit: parity tests, bench, rocprofv3 kernel trace + PMC passes (outputs under gpurun_out/)
#   tools/gpu_round.sh TAG [quick]      quick = skip pytest / smoke / default bench (profiles only)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-r02}
if [ "$2" != "quick" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log
BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 48 --warmup 4 --batches 4 --no-cpu > gpurun_out/bench_2rank_gloo.log 2>&1; echo "2-rank gloo rehearsal rc=$?"
tail -1 gpurun_out/bench_2rank_gloo.log
fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o trace -- python bench.py --steps 48 --warmup 2 --no-cpu > gpurun_out/rocprof_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/${TAG}_fetch -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/rocprof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/${TAG}_write -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/rocprof_write.log 2>&1; echo "write rc=$?"
bash tools/pmc.sh $TAG > gpurun_out/pmc_${TAG}.txt 2>&1; tail -30 gpurun_out/pmc_${TAG}.txt
timeout 120 python tools/phase_prof.py > gpurun_out/phase.log 2>&1; tail -1 gpurun_out/phase.log
find gpurun_out/prof -name "*.csv" | head -20
