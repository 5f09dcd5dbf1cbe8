#!/bin/bash
# one GPU-box visit: parity tests, bench, rocprofv3 kernel trace + PMC passes (outputs under gpurun_out/)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log
TAG=${1:-r01}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o trace -- python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/rocprof_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/${TAG}_fetch -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/rocprof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/${TAG}_write -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/rocprof_write.log 2>&1; echo "write rc=$?"
timeout 120 python tools/phase_prof.py > gpurun_out/phase.log 2>&1; tail -1 gpurun_out/phase.log
find gpurun_out/prof -name "*.csv" | head -20
