cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r02i.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r02i.log
timeout 900 python bench.py > gpurun_out/bench_r02i.log 2>&1; tail -1 gpurun_out/bench_r02i.log
