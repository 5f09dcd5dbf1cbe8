#!/bin/bash
# round 5, third GPU visit: -f / UMI on the lane plan and --dedup without the hash pre-pass on the hardware (goldens, oracle
# parity at scale for configs[4], the option fuzz), then the bench line with the other configurations.
#   gpurun --timeout 1200 -- 'bash tools/gpu_r5_d.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider -k "umi or trim_fixed or dedup or config4 or config5 or plans_agree or random_option or cells_at or at_baseline_scale or two_shard" > gpurun_out/r5d_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5d_pytest.log
timeout 500 python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu > gpurun_out/r5d_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r5d_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], 'Mreads/s', j['ms_per_step'], 'ms/step', j['roofline'])
for r in j.get('other_configs') or []: print(json.dumps(r))
"
rm -rf gpurun_out/prof/r5d_cfg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r5d_cfg -o t -- python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0)): print(json.dumps(r))
" > gpurun_out/r5d_other_configs.log 2>&1; echo "other configs trace rc=$?"
python - > gpurun_out/r5d_other_configs_kernels.txt <<'PY'
import csv, glob
print("per-kernel times of bench.other_configs() under rocprofv3 --kernel-trace --stats (all configurations in one process, in order)")
f = glob.glob("gpurun_out/prof/r5d_cfg/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "fq_" in r["Name"]:
        print(f"{r['Name'][:70]:72s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:8.4f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
cat gpurun_out/r5d_other_configs_kernels.txt
find gpurun_out/prof -name "*_kernel_trace.csv" -delete
