#!/bin/bash
# round 5: read 2's quality rows by global_load_lds under read 1's hash (FASTP_GPU_LANE_GLDS A/B, kernel averages), parity of
# that path, then the other configurations with a kernel trace (where -c's 6.8 ms go)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
OUT=gpurun_out/r5h_lane_glds.txt
: > $OUT
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "plans_agree or at_baseline_scale or correction or random_option or trim_fixed or umi or equals_oracle_at_scale" > gpurun_out/r5h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5h_pytest.log
trace() {
  NAME=$1; shift
  rm -rf gpurun_out/prof/r5h_$NAME
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r5h_$NAME -o t -- python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/r5h_$NAME.log 2>&1
  python - "$NAME" "$@" >> $OUT <<'PY'
import csv, glob, sys
name = sys.argv[1]
f = glob.glob(f"gpurun_out/prof/r5h_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(sys.argv[2:])}]"
if f:
    for r in csv.DictReader(open(f[0])):
        for key in ("fq_lane_kernel", "fq_stats_kernel"):
            if key in r["Name"]: line += f"  {key} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
print(line)
PY
  find gpurun_out/prof/r5h_$NAME -name "*_kernel_trace.csv" -delete
}
trace glds_on   FASTP_GPU_LANE_GLDS=1
trace glds_off  FASTP_GPU_LANE_GLDS=0
trace glds_on_b FASTP_GPU_LANE_GLDS=1
trace glds_off_b FASTP_GPU_LANE_GLDS=0
cat $OUT
rm -rf gpurun_out/prof/r5h_cfg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r5h_cfg -o t -- python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0)): print(json.dumps(r))
" > gpurun_out/r5h_other_configs.log 2>&1; echo "other configs trace rc=$?"
grep '^{' gpurun_out/r5h_other_configs.log | cut -c1-260
python - > gpurun_out/r5h_other_configs_kernels.txt <<'PY'
import csv, glob
print("per-kernel times of bench.other_configs() under rocprofv3 --kernel-trace --stats (all configurations in one process, in order)")
f = glob.glob("gpurun_out/prof/r5h_cfg/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "fq_" in r["Name"]:
        print(f"{r['Name'][:70]:72s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:8.4f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
cat gpurun_out/r5h_other_configs_kernels.txt
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
