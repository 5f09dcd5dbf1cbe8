#!/bin/bash
# the other configurations (bench.py's other_configs leg) + overrep parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overrep or config4 or config1" > gpurun_out/pytest_cfg.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_cfg.log
timeout 600 python bench.py --steps 8 --warmup 2 --batches 4 --no-cpu > gpurun_out/cfg.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/cfg.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps(j.get('other_configs'), indent=1))"
