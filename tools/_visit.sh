cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_final.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_final.log | cut -c1-200
