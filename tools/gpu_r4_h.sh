#!/bin/bash
# round 4: the text kernel with its Stats tables in LDS: parity again, and what it costs now
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "exotic or text_kernel" > gpurun_out/pytest_exotic3.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_exotic3.log
timeout 300 python tools/exotic_bench.py > gpurun_out/r04_exotic_cost_lds.txt 2>&1; echo "bench rc=$?"; cat gpurun_out/r04_exotic_cost_lds.txt
FASTP_GPU_EXACT_LDS=0 timeout 300 python tools/exotic_bench.py 100000 > gpurun_out/r04_exotic_cost_nolds_100k.txt 2>&1; tail -3 gpurun_out/r04_exotic_cost_nolds_100k.txt
timeout 300 python tools/exotic_bench.py 100000 > gpurun_out/r04_exotic_cost_lds_100k.txt 2>&1; tail -3 gpurun_out/r04_exotic_cost_lds_100k.txt
