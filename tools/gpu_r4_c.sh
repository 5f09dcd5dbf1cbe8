#!/bin/bash
# round 4, third GPU visit: lane-kernel metrics A/B on ONE box (rocprofv3 per-kernel times), the extended lane plan on the other
# configurations, the drop-in with the faster start-up
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
for V in m0 m2; do
  L=$GRAFT_REPO_ROOT/fastp_amd/libfastp_gpu.so; [ $V = m0 ] && L=$GRAFT_REPO_ROOT/fastp_amd/libfastp_gpu_m0.so
  FASTP_GPU_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r04_$V -o trace -- python bench.py --steps 48 --warmup 2 --no-cpu --no-extras > gpurun_out/rocprof_$V.log 2>&1; echo "trace $V rc=$?"
  tail -1 gpurun_out/rocprof_$V.log | cut -c1-200
  python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/prof/r04_$V/trace_kernel_stats.csv")))
for r in rows:
    if "fq_lane" in r["Name"] or r["Name"] in ("fq_stats_kernel",):
        print("$V", r["Name"][:60], "calls", r["Calls"], "avg_ms", round(float(r["AverageNs"]) / 1e6, 4))
PY
done > gpurun_out/r04_lane_metrics_ab.txt 2>&1
cat gpurun_out/r04_lane_metrics_ab.txt
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
timeout 400 python - > gpurun_out/r04_other_configs.txt 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench
for r in bench.other_configs(torch.device("cuda", 0)):
    print(json.dumps(r))
PY
cat gpurun_out/r04_other_configs.txt
timeout 500 python tools/dropin_bench.py --pairs 4000000 --big 12000000 --quick > gpurun_out/r04_dropin2.txt 2>&1; echo "dropin rc=$?"
cat gpurun_out/r04_dropin2.txt | cut -c1-420
