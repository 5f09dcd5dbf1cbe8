#!/bin/bash
# round 6, visit r: where the patched reference's wall clock goes (start-up timeline, the file loop under a few I/O settings),
# and what FETCH_SIZE reports per load width on a known byte count (tools/microbench/fetch_calib.hip)
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_r.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6r}
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/${V}_calib -o pmc -- tools/microbench/fetch_calib > gpurun_out/${V}_fetch_calib.txt 2>&1; echo "calib rc=$?"
python - >> gpurun_out/${V}_fetch_calib.txt <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof/${V}_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE":
            print(f"{r['Kernel_Name'][:60]:62s} FETCH_SIZE {float(r['Counter_Value']):14.0f} KiB = {float(r['Counter_Value']) * 1024 / (1 << 30):.4f} of the 1 GiB read")
PY
cat gpurun_out/${V}_fetch_calib.txt
timeout 1200 python tools/dropin_probe.py 4000000 12000000 > gpurun_out/${V}_dropin_probe.txt 2> gpurun_out/${V}_dropin_probe.err; echo "probe rc=$?"
cat gpurun_out/${V}_dropin_probe.txt
tail -5 gpurun_out/${V}_dropin_probe.err
