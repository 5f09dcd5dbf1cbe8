#!/bin/bash
# round 6, visit j: fq_ovr_count_kernel with DS addresses (configs[4]) - the overrepresentation cases on the hardware, the
# configuration's kernels, its line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=r6j
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "overrep or config4" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
rm -rf gpurun_out/prof/${V}_cfg4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_cfg4 -o t -- python tools/one_config.py "configs[4]" > gpurun_out/${V}_cfg4.log 2>&1
python - <<'PY' | tee gpurun_out/r6j_cfg4_kernels.txt
import csv, glob
f = glob.glob("gpurun_out/prof/r6j_cfg4/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "at::native" not in r["Name"] and "elementwise" not in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    n = r["Name"].split("(")[0].replace("void fq::", "").replace("void ", "")[:60]
    print(f"   {n:60s} avg {float(r['AverageNs'])/1e6:8.4f} ms  x{r['Calls']:>4s}")
PY
grep '^{' gpurun_out/${V}_cfg4.log | cut -c1-260 | tee -a gpurun_out/r6j_cfg4_kernels.txt
find gpurun_out/prof/${V}_cfg4 -name "*_kernel_trace.csv" -delete
timeout 200 python tools/one_config.py "configs[4]" | cut -c1-260 | tee -a gpurun_out/r6j_cfg4_kernels.txt
