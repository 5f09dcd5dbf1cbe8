#!/bin/bash
# kernel trace of the other configurations (bench.py's other_configs leg) + the full default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/cfg -o trace -- python bench.py --steps 8 --warmup 2 --batches 4 --no-cpu > gpurun_out/cfg_trace.log 2>&1; echo "trace rc=$?"
tail -1 gpurun_out/cfg_trace.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps(j.get('other_configs'), indent=1))"
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof/cfg/trace_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if n.startswith(("fq_", "void fq", "__amd")):
        print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
PY
