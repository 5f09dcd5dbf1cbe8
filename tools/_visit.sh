cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
./tools/microbench/issue_rate > gpurun_out/issue_rate.txt 2>&1; echo "microbench rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r02c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r02c.log
timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu > gpurun_out/bench_r02c.log 2>&1; tail -1 gpurun_out/bench_r02c.log
timeout 120 python tools/phase_prof.py > gpurun_out/phase_r02c.log 2>&1; tail -1 gpurun_out/phase_r02c.log
