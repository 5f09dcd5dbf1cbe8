cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { "$@" timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
echo "claim issued in the trim phase by idle waves"; run env
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "equals_oracle or stream or launches or baseline_scale or reset" > gpurun_out/pytest_r02s.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r02s.log
