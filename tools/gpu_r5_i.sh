#!/bin/bash
# round 5: -c after the Stats fix-up gathers in LDS: parity, then the other configurations with a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "correction or corr or random_option or plans_agree or overrep or config4 or config5" > gpurun_out/r5i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5i_pytest.log
rm -rf gpurun_out/prof/r5i_cfg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r5i_cfg -o t -- python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0)): print(json.dumps(r))
" > gpurun_out/r5i_other_configs.log 2>&1; echo "other configs trace rc=$?"
grep '^{' gpurun_out/r5i_other_configs.log | cut -c1-260
python - > gpurun_out/r5i_other_configs_kernels.txt <<'PY'
import csv, glob
print("per-kernel times of bench.other_configs() under rocprofv3 --kernel-trace --stats (all configurations in one process, in order)")
f = glob.glob("gpurun_out/prof/r5i_cfg/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "fq_" in r["Name"]:
        print(f"{r['Name'][:70]:72s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:8.4f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
cat gpurun_out/r5i_other_configs_kernels.txt
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
