cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r02t -o trace -- python bench.py --steps 48 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
grep "^fq_\|^\"fq_" gpurun_out/prof/r02t/trace_kernel_stats.csv | cut -c1-110
