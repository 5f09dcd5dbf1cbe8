#!/bin/bash
# round 6, visit s: the drop-in with the buffers allocated beside the engine's creation and the fast exit (timeline + walls), the
# worker loop's step as a timeline of its kernels, configs[4] with fq_ovr_tasks_kernel's slots taken per workgroup
#   gpurun --timeout 1800 -- 'bash tools/gpu_r6_s.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6s}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "overrep or config4 or stream or patched_reference" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
timeout 600 python tools/dropin_probe.py 4000000 12000000 > gpurun_out/${V}_dropin_probe.txt 2> gpurun_out/${V}_dropin_probe.err; echo "probe rc=$?"
cat gpurun_out/${V}_dropin_probe.txt | cut -c1-330
rm -rf gpurun_out/prof/${V}_step
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/${V}_step -o t -- python bench.py --steps 24 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/${V}_step.log 2>&1; echo "step trace rc=$?"
python tools/step_timeline.py gpurun_out/prof/${V}_step 3 > gpurun_out/${V}_step_timeline.txt 2>&1
cat gpurun_out/${V}_step_timeline.txt
find gpurun_out/prof/${V}_step -name "*_kernel_trace.csv" -delete
rm -rf gpurun_out/prof/${V}_c4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_c4 -o t -- python tools/one_config.py "configs[4]" > gpurun_out/${V}_c4.log 2>&1; echo "configs[4] rc=$?"
grep '^{' gpurun_out/${V}_c4.log | cut -c1-300
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof/${V}_c4/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "fq_" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f"   {r['Name'].split('(')[0][:50]:52s} {float(r['AverageNs'])/1e6:8.4f} ms x{r['Calls']}")
PY
find gpurun_out/prof/${V}_c4 -name "*_kernel_trace.csv" -delete
