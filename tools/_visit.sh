cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof; export TMPDIR=/tmp
run() { "$@" timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
echo "default"; run env
echo "HALVES=2"; run env FASTP_GPU_HALVES=2
echo "HALVES=2 skew 3"; run env FASTP_GPU_HALVES=2 FASTP_GPU_HALF_SKEW=3
echo "TILE=96"; run env FASTP_GPU_TILE=96
echo "TILE=112"; run env FASTP_GPU_TILE=112
echo "TILE=120"; run env FASTP_GPU_TILE=120
timeout 600 python tools/aux_bench.py 400000 2>&1 | grep -i "deflate\|ratio"
timeout 600 python tools/aux_bench.py 2000000 2>&1 | grep -i "deflate\|ratio"
