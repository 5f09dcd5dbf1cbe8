#!/bin/bash
# one GPU-box visit, round 6: full -m gpu suite, smoke, the driver's bench command, the default bench (with the e2e legs), kernel
# trace, FETCH / WRITE and SQ counter passes, condensed on the box into gpurun_out/summary_<tag>/ (copied into profiles/ afterwards).
#   tools/gpu_round5.sh TAG [notests]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-r06}
if [ "$2" != "notests" ]; then
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2>&1; echo "bench (driver's command) rc=$?"; tail -1 gpurun_out/bench_driver_cmd.log | cut -c1-2500
timeout 900 python bench.py --no-extras > gpurun_out/bench.log 2>&1; echo "bench (default steps, no extras) rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-700
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o trace -- python bench.py --steps 48 --warmup 2 --no-cpu --no-extras > gpurun_out/rocprof_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/${TAG}_fetch -o pmc -- python bench.py --steps 2 --warmup 1 --batches 2 --no-cpu --no-extras > gpurun_out/rocprof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/${TAG}_write -o pmc -- python bench.py --steps 2 --warmup 1 --batches 2 --no-cpu --no-extras > gpurun_out/rocprof_write.log 2>&1; echo "write rc=$?"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${TAG}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu --no-extras > gpurun_out/pmc_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
# per-kernel breakdown of the other configurations (configs[1], the SE default with its adapter, PE with adapter sequences, configs[4]'s share)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${TAG}_cfg -o trace -- python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0)): print(json.dumps(r))
" > gpurun_out/other_configs.log 2>&1; echo "other configs trace rc=$?"
grep '^{' gpurun_out/other_configs.log
mkdir -p gpurun_out/summary_$TAG
SUMMARIZE_DST=gpurun_out/summary_$TAG python tools/summarize_prof.py $TAG 4194304 > /dev/null 2> gpurun_out/summarize.err; echo "summarize rc=$?"; tail -2 gpurun_out/summarize.err
for K in fq_lane_kernel fq_stats5_kernel; do echo "== SQ counters, $K (one launch of 4194304 pairs)"; python tools/pmc_parse.py $TAG $K; done > gpurun_out/summary_$TAG/${TAG}_sq_counters.txt
python - > gpurun_out/summary_$TAG/${TAG}_other_configs_kernels.txt <<PY
import csv
print("per-kernel times of bench.other_configs() under rocprofv3 --kernel-trace --stats (all four configurations in one process, in order)")
for r in csv.DictReader(open("gpurun_out/prof/${TAG}_cfg/trace_kernel_stats.csv")):
    if r["Name"].startswith("fq_") or "fq_lane" in r["Name"]:
        print(f"{r['Name'][:70]:72s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:8.4f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
find gpurun_out/prof -name "*counter_collection.csv" -size +2M -delete
du -sh gpurun_out
cat gpurun_out/summary_$TAG/${TAG}_sq_counters.txt | head -50
cat gpurun_out/summary_$TAG/traffic.json | head -12
