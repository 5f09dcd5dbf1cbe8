#!/bin/bash
# round 6, visit m: the correction list entries of a chunk taken with one atomic (lane_flush_corrections), Stats form 5 back to
# visit f's form, what letters outside ACGTN cost with the text kernel as a wavefront per unit, the driver's command
#   gpurun --timeout 2400 -- 'bash tools/gpu_r6_m.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6m}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "corr or merge or exotic or text_kernel or plans_agree or baseline_scale or stats" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
summ() {  # V NAME ARGS...
  python - "$@" >> $OUT <<'PY'
import csv, glob, sys, json
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(a for a in sys.argv[3:] if 'FASTP_GPU_LIB' not in a and 'BENCH_ALLOW' not in a)}{' (profiling build)' if any('FASTP_GPU_LIB' in a for a in sys.argv[3:]) else ''}]"
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "at::native" not in r["Name"] and "elementwise" not in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        n = r["Name"].split("(")[0].replace("void fq::", "").replace("void ", "")[:44]
        line += f"  {n} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
print(line)
PY
  tail -1 $OUT | cut -c1-400
}
trace() {   # NAME ENV... : kernel averages of the bench's configuration
  NAME=$1; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/${V}_$NAME.log 2>&1
  summ "$V" "$NAME" "$@"
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
}
tracec() {   # NAME CONFIG ENV... : kernel averages of one line of other_configs
  NAME=$1; CFG=$2; shift; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python tools/one_config.py "$CFG" > gpurun_out/${V}_$NAME.log 2>&1
  summ "$V" "$NAME" "$@"
  grep '^{' gpurun_out/${V}_$NAME.log | cut -c1-200 >> $OUT
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
}
trace  headline           FASTP_GPU_VERBOSE=1
tracec c_line             " -c "  FASTP_GPU_VERBOSE=1
tracec merge_line         "--merge"  FASTP_GPU_VERBOSE=1
tracec softmask           "soft-masked"  FASTP_GPU_VERBOSE=1
cat $OUT
timeout 300 python tools/exotic_bench.py > gpurun_out/${V}_exotic_cost.txt 2>&1; cat gpurun_out/${V}_exotic_cost.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${V}_bench_driver_cmd.log 2> gpurun_out/${V}_bench_driver_cmd.err; echo "bench rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log > gpurun_out/${V}_bench_driver_cmd.json; python - "$V" <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_driver_cmd.json").read())
print(j["value"], j["ms_per_step"], j["roofline"])
for r in j.get("other_configs", []): print({k: v for k, v in r.items() if k in ("config", "ms_per_step", "plan", "frac", "error")})
for k in ("cpu_baseline", "e2e_gpu", "e2e_dropin", "e2e_dropin_large", "e2e_dropin_bgzf"): print(k, {a: b for a, b in j.get(k, {}).items() if a not in ("what", "sample", "plain_gzip_inputs")})
PY
