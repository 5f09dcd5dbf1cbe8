cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for S in "31" "15,31" "23,31" "7,15,23,31" "3,11,19,27" "30,31" "0,1,2,3"; do
  echo "AUX_SLOTS=$S"; FASTP_GPU_AUX_SLOTS=$S timeout 600 python bench.py --steps 48 --warmup 8 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done > gpurun_out/aux_sweep2.log 2>&1
cat gpurun_out/aux_sweep2.log
