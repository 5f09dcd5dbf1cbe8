#!/bin/bash
# one GPU-box visit, round 3: full -m gpu suite, smoke, default bench (with the e2e legs), kernel trace, FETCH / WRITE
# and SQ counter passes.   tools/gpu_round3.sh TAG [quick]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-r03}
if [ "$2" = "profile" ]; then SKIP_BENCH=1; fi
if [ "$2" != "quick" ] && [ "$2" != "profile" ]; then
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
fi
if [ -z "$SKIP_BENCH" ]; then
timeout 1200 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log | cut -c1-1500
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/bench_steps20.log 2>&1; echo "bench --steps 20 rc=$?"; tail -1 gpurun_out/bench_steps20.log | cut -c1-600
fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o trace -- python bench.py --steps 48 --warmup 2 --no-cpu --no-extras > gpurun_out/rocprof_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/${TAG}_fetch -o pmc -- python bench.py --steps 2 --warmup 1 --batches 2 --no-cpu --no-extras > gpurun_out/rocprof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/${TAG}_write -o pmc -- python bench.py --steps 2 --warmup 1 --batches 2 --no-cpu --no-extras > gpurun_out/rocprof_write.log 2>&1; echo "write rc=$?"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
P3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${TAG}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu --no-extras > gpurun_out/pmc_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
# keep what comes back small: the per-dispatch trace rows are not needed once the stats file exists
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
du -sh gpurun_out
for K in fq_lane_kernel fq_stats_kernel; do echo "== SQ counters, $K (one launch of 4194304 pairs)"; python tools/pmc_parse.py $TAG $K; done > gpurun_out/sq_${TAG}.txt
cat gpurun_out/sq_${TAG}.txt | head -60
