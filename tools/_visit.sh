cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { "$@" timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
echo "claim/winners/finish"; run env
echo "table (probe/resolve)"; run env FASTP_GPU_DUP_TABLE=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_option_fuzz.py -m gpu -x -q -k "equals_oracle or dedup or stream or launches or shard or baseline_scale or bit_positions or fuzz or config5 or reset" > gpurun_out/pytest_r02q.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r02q.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r02q -o trace -- python bench.py --steps 24 --warmup 2 --no-cpu > /dev/null 2>&1; grep "^fq_\|^\"fq_" gpurun_out/prof/r02q/trace_kernel_stats.csv | cut -c1-110
