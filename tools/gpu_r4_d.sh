#!/bin/bash
# round 4, fourth GPU visit: lane kernel geometry A/B on one box (workgroup size x workgroups per CU), per-kernel times from rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
for T in 768 384 256 512; do
  FASTP_GPU_LANE_THREADS=$T FASTP_GPU_VERBOSE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r04_t$T -o trace -- python bench.py --steps 48 --warmup 2 --no-cpu --no-extras > gpurun_out/rocprof_t$T.log 2>&1; echo "trace threads=$T rc=$?"
  grep "lane kernel" gpurun_out/rocprof_t$T.log | head -1
  tail -1 gpurun_out/rocprof_t$T.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  value', j['value'], 'ms_per_step', j['ms_per_step'], 'kernel_avg_ms', j['roofline']['kernel_avg_ms'])"
  python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof/r04_t$T/trace_kernel_stats.csv")):
    if "fq_lane" in r["Name"] or r["Name"] == "fq_stats_kernel":
        print("  threads=$T", r["Name"][:58], "calls", r["Calls"], "avg_ms", round(float(r["AverageNs"]) / 1e6, 4))
PY
done > gpurun_out/r04_lane_geometry_ab.txt 2>&1
cat gpurun_out/r04_lane_geometry_ab.txt
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
timeout 400 python - > gpurun_out/r04_other_configs.txt 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench
for r in bench.other_configs(torch.device("cuda", 0)):
    print(json.dumps(r))
PY
cat gpurun_out/r04_other_configs.txt
